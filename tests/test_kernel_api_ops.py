"""The kernel-package boundary as torch sees it (VERDICT r4 missing #2 / weak #4, CPU — no kernel runs):
  * the public entry points have the reference's positional arguments and defaults (fastvideo_kernel/ops.py:21-28, 65-74, 136-145),
    checked against the reference's own source when it is present and against the same list written down here when it is not;
  * every kernel call sits inside a torch.library op with a fake kernel, so a caller traces under torch.compile(fullgraph=True) on fake
    "cuda" tensors without a graph break (the reference registers its kernels the same way: block_sparse_attn.py:103-145, 224-267);
  * there is no CPU kernel behind those ops (no fallback): a CPU tensor is refused by the dispatcher;
  * the text-token extension of the sliding-tile block lists is the reference's mask (support_flex_sta.py:52-55), bit for bit."""
import ast
import inspect
import os

import numpy as np
import pytest
import torch

from fastvideo_amd import kernel_api as KA

REF_OPS = "/root/reference/fastvideo-kernel/python/fastvideo_kernel/ops.py"
_REQ = inspect.Parameter.empty
# (name, default) in order — fastvideo_kernel/ops.py:21-28, 65-74, 136-145
EXPECTED = {
    "sliding_tile_attention": [("q", _REQ), ("k", _REQ), ("v", _REQ), ("window_size", _REQ), ("text_length", _REQ), ("has_text", True),
                               ("seq_shape", "30x48x80")],
    "video_sparse_attn": [("q", _REQ), ("k", _REQ), ("v", _REQ), ("variable_block_sizes", _REQ), ("q_variable_block_sizes", _REQ),
                          ("topk", _REQ), ("block_size", 64), ("compress_attn_weight", None)],
    "video_sparse_attn_bshd": [("q", _REQ), ("k", _REQ), ("v", _REQ), ("variable_block_sizes", _REQ), ("q_variable_block_sizes", _REQ),
                               ("topk", _REQ), ("block_size", 64), ("compress_attn_weight", None)],
}


def _ours(name):
    return [(p.name, p.default) for p in inspect.signature(getattr(KA, name)).parameters.values()]


@pytest.mark.parametrize("name", sorted(EXPECTED))
def test_entry_points_take_the_reference_arguments_and_defaults(name):
    ours = _ours(name)
    exp = EXPECTED[name]
    assert ours[:len(exp)] == exp, f"{name}: {ours[:len(exp)]} != reference {exp}"
    assert all(d is not _REQ for _, d in ours[len(exp):]), "extra parameters must be optional"


@pytest.mark.skipif(not os.path.exists(REF_OPS), reason="reference source not present (GPU box)")
def test_expected_signatures_are_the_reference_source():
    tree = ast.parse(open(REF_OPS).read())
    seen = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in EXPECTED:
            args = [a.arg for a in node.args.args]
            defaults = [_REQ] * (len(args) - len(node.args.defaults)) + [ast.literal_eval(d) for d in node.args.defaults]
            seen[node.name] = list(zip(args, defaults))
    assert seen == EXPECTED


def test_callers_trace_under_torch_compile_fullgraph_on_fake_tensors():
    from torch._dynamo.testing import CompileCounterWithBackend
    from torch._subclasses.fake_tensor import FakeTensorMode

    def caller(q, k, v, vbs, idx, num):
        a = KA.video_sparse_attn(q * 2, k, v, vbs, vbs, 3, (4, 4, 4), compress_attn_weight=v)
        b = KA.sliding_tile_attention(a, k, v, [(1, 1, 1), (3, 1, 1)], 0, False, "6x8x16")
        o, lse = KA.block_sparse_attn_from_indices(b, k, v, idx, num, vbs)
        c = KA.video_sparse_attn_bshd(o.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), vbs[:6], vbs[:6], 2, 128)
        return c + 1, lse

    with FakeTensorMode():
        q = torch.empty((1, 2, 768, 128), dtype=torch.bfloat16, device="cuda")
        vbs = torch.full((12,), 64, dtype=torch.int32, device="cuda")
        idx = torch.empty((1, 2, 12, 12), dtype=torch.int32, device="cuda")
        num = torch.empty((1, 2, 12), dtype=torch.int32, device="cuda")
        cnt = CompileCounterWithBackend("eager")
        out, lse = torch.compile(caller, backend=cnt, fullgraph=True)(q, q, q, vbs, idx, num)
    assert cnt.frame_count == 1, "graph break"
    assert tuple(out.shape) == (1, 768, 2, 128) and out.dtype == torch.bfloat16 and tuple(lse.shape) == (1, 2, 768) and lse.dtype == torch.float32
    targets = [str(n.target) for g in cnt.graphs for n in g.graph.nodes if n.op == "call_function"]
    for op in ("video_sparse_attn_gfx950", "sliding_tile_attention_gfx950", "block_sparse_attn_gfx950"):
        assert any(op in t for t in targets), (op, targets)


def test_ops_have_no_cpu_kernel():
    q = torch.zeros((1, 1, 768, 128), dtype=torch.bfloat16)
    vbs = torch.full((12,), 64, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        KA.video_sparse_attn(q, q, q, vbs, vbs, 3, (4, 4, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        KA.sliding_tile_attention(q, q, q, [(1, 1, 1)], 0, False, "6x8x16")
    # and the ops themselves have no CPU kernel behind the wrappers' check: the dispatcher refuses
    with pytest.raises((NotImplementedError, RuntimeError), match="CPU"):
        torch.ops.fastvideo_kernel.video_sparse_attn_gfx950(q, q, q, vbs, vbs, 3, 64, None, "bhsd")


def test_argument_errors_come_before_any_kernel():
    q = torch.zeros((1, 2, 768 + 100, 128), dtype=torch.bfloat16)
    with pytest.raises(ValueError, match="has_text=False"):
        KA.sliding_tile_attention(q, q, q, [(1, 1, 1)] * 2, 0, False, "6x8x16")
    with pytest.raises(ValueError, match="text tokens"):
        KA.sliding_tile_attention(q, q, q, [(1, 1, 1)] * 2, 101, True, "6x8x16")       # more valid text tokens than text rows
    with pytest.raises(ValueError, match="one \\(t,h,w\\) per head"):
        KA.sliding_tile_attention(q, q, q, [(1, 1, 1)], 10, True, "6x8x16")
    with pytest.raises(ValueError, match="not divisible"):
        KA.sliding_tile_attention(q, q, q, [(1, 1, 1)] * 2, 10, True, "7x8x16")
    with pytest.raises(RuntimeError, match="bf16 only"):
        KA.sliding_tile_attention(q.float(), q.float(), q.float(), [(1, 1, 1)] * 2, 10, True, "6x8x16")


@pytest.mark.parametrize("text_rows,text_length", [(384, 0), (384, 1), (384, 100), (384, 256), (768, 385)])
def test_text_block_lists_are_the_reference_mask(text_rows, text_length):
    """_canvas_tile_lists with text rows, expanded to a dense [Sq, Skv] boolean mask, == oracle sta_mask (support_flex_sta.py:29-59) with
    the same text_length on the padded length — integer work, bit-exact."""
    from oracle import vsa_oracle as V
    canvas, tile, tok = (12, 16, 24), (6, 8, 8), 384
    tiles = tuple(c // t for c, t in zip(canvas, tile))
    img = int(np.prod(canvas))
    windows = ((3, 3, 3), (1, 1, 3), (3, 1, 1))
    idx, num, sizes = KA._canvas_tile_lists(tiles, tok, windows, 1, "cpu", text_rows=text_rows, text_length=text_length)
    S = img + text_rows
    assert sizes.numel() == S // 64 and idx.shape[:3] == (1, 3, S // tok)
    for h, w in enumerate(windows):
        m = torch.zeros((S, S), dtype=torch.bool)
        for t in range(S // tok):
            for b in idx[0, h, t, :int(num[0, h, t])].tolist():
                m[t * tok:(t + 1) * tok, b * 64:b * 64 + int(sizes[b])] = True
            lst = idx[0, h, t, :int(num[0, h, t])]
            assert (lst[1:] > lst[:-1]).all(), "lists ascend"
        ref = V.sta_mask(canvas, w, tile, text_length=text_length, total_len=S)
        assert torch.equal(m, ref), (w, int((m != ref).sum()))
