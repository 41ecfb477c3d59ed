"""torch.Tensor front-ends of the C ABI (include/fvk_amd.h).  PyTorch is plumbing here: device memory,
the current HIP stream, and output allocation; every computation below is a hand-written gfx950 kernel in
libfvk_amd.so.  All functions require ROCm tensors and raise on anything else — no eager fallback."""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib
from ._lib import AttnArgs

BF16 = torch.bfloat16
EPI_NONE, EPI_GELU_TANH, EPI_SILU, EPI_RESIDUAL_GATE, EPI_DIV = 0, 1, 2, 3, 4
LN_ROUND_RESIDUAL, LN_ROUND_NORM = 1, 2


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t: torch.Tensor, dtype, name: str):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"fastvideo_amd: {name} must be a ROCm device tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise RuntimeError(f"fastvideo_amd: {name} must be {dtype}, got {t.dtype}")
    if t.device.index != torch.cuda.current_device():
        # launches go to torch.cuda.current_stream() of the CURRENT device: a tensor of another GPU would be dereferenced there
        raise RuntimeError(f"fastvideo_amd: {name} lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                           "wrap the call in torch.cuda.device(tensor.device)")
    return t


def _f32(t, name):
    if t is None:
        return None
    return _chk(t, torch.float32, name).contiguous()


def set_tunable(name: str, value: int) -> None:
    """Measurement switches of the library (include/fvk_amd.h fvk_set_tunable); 0 = shipped configuration."""
    _lib.call("fvk_set_tunable", name.encode(), int(value))


# ------------------------------------------------------------------ norm / modulate
def ln_modulate(x, *, residual=None, gate=None, ln_w=None, ln_b=None, mul=None, add=None, eps=1e-6,
                round_residual=False, round_norm=False, want_residual=False, rows_per_batch=None, fp8_rowwise=None):
    """See fvk_ln_modulate_bf16.  x/residual bf16 [..., d]; gate/mul/add fp32 [B, d] (or [B,1,d]).
    fp8_rowwise: None = bf16 output only; "only" = the per-token e4m3 quantisation (q [M,d] float8_e4m3fn, scale [M,1] fp32) INSTEAD of the
    bf16 output, "both" = the bf16 output and the pair (fvk_ln_modulate_fp8_bf16).  The first returned value is then (q, scale) or
    (out, (q, scale))."""
    _chk(x, BF16, "x")
    d = x.shape[-1]
    x2 = x.contiguous().view(-1, d)
    M = x2.shape[0]
    res2 = None
    if residual is not None:
        res2 = _chk(residual, BF16, "residual").contiguous().view(-1, d)
    B = 1
    for t in (gate, mul, add):
        if t is not None:
            B = max(B, t.numel() // d)
    if rows_per_batch is None:
        rows_per_batch = max(M // B, 1)
    if fp8_rowwise not in (None, "only", "both"):
        raise ValueError(f"ln_modulate: fp8_rowwise={fp8_rowwise!r} (None, 'only' or 'both')")
    out = torch.empty_like(x2) if fp8_rowwise != "only" else None
    res_out = torch.empty_like(x2) if want_residual else None
    flags = (LN_ROUND_RESIDUAL if round_residual else 0) | (LN_ROUND_NORM if round_norm else 0)
    gate, mul, add, ln_w, ln_b = (_f32(t, n) for t, n in ((gate, "gate"), (mul, "mul"), (add, "add"),
                                                          (ln_w, "ln_w"), (ln_b, "ln_b")))
    if fp8_rowwise is None:
        _lib.call("fvk_ln_modulate_bf16", _p(x2), _p(res2), _p(gate), _p(ln_w), _p(ln_b), _p(mul), _p(add), _p(res_out),
                  _p(out), M, d, rows_per_batch, float(eps), flags, _stream())
        first = out.view(x.shape)
    else:
        q = torch.empty((M, d), dtype=FP8, device=x.device)
        qs = torch.empty((M,), dtype=torch.float32, device=x.device)
        _lib.call("fvk_ln_modulate_fp8_bf16", _p(x2), _p(res2), _p(gate), _p(ln_w), _p(ln_b), _p(mul), _p(add), _p(res_out),
                  _p(out), _p(q), _p(qs), M, d, rows_per_batch, float(eps), flags, _stream())
        first = (q, qs.view(M, 1)) if out is None else (out.view(x.shape), (q, qs.view(M, 1)))
    return (first, res_out.view(x.shape)) if want_residual else first


def scale_residual(residual, x, gate=None, rows_per_batch=None):
    _chk(x, BF16, "x"), _chk(residual, BF16, "residual")
    d = x.shape[-1]
    x2, r2 = x.contiguous().view(-1, d), residual.contiguous().view(-1, d)
    M = x2.shape[0]
    gate = _f32(gate, "gate")
    B = 1 if gate is None else gate.numel() // d
    out = torch.empty_like(x2)
    _lib.call("fvk_scale_residual_bf16", _p(r2), _p(x2), _p(gate), _p(out), M, d, rows_per_batch or max(M // B, 1), _stream())
    return out.view(x.shape)


import weakref

_ROW_MAPS_OK: dict = {}  # id(map tensor) -> {(version, output rows)}; the entry dies with the tensor (weakref.finalize)


def _check_row_map(rmap, n_out_rows):
    """A scatter map is dereferenced by the kernel without a bound: check max(map) < rows of the output ONCE per (map tensor,
    version, output rows) — the maps are per-geometry metadata built once and reused every layer, so this costs one device
    sync per geometry, not per call.  Keyed on the tensor OBJECT, not its address: the entry is dropped when the tensor is collected, so
    a freed map whose storage (or id) is reused by another map of the same length can never inherit the verdict."""
    seen = _ROW_MAPS_OK.get(id(rmap))
    key = (rmap._version, int(n_out_rows))
    if seen is not None and key in seen:
        return
    hi = int(rmap.max().item()) if rmap.numel() else -1
    if hi >= n_out_rows:
        raise RuntimeError(f"rmsnorm_rope: row map addresses row {hi} of an output with {n_out_rows} rows")
    if seen is None:
        _ROW_MAPS_OK[id(rmap)] = {key}
        weakref.finalize(rmap, _ROW_MAPS_OK.pop, id(rmap), None)
    else:
        seen.add(key)


def rmsnorm_rope(tensors, weights=None, cos=None, sin=None, head_dim=128, seq_len=None, eps=1e-6, outs=None, pos_offset=0, row_maps=None):
    """tensors: list (<=3) of bf16 2-D views [M, width] sharing one row stride (e.g. q,k column slices of a fused
    QKV buffer).  Returns a list of new contiguous [M, width] tensors (or writes `outs`).
    row_maps (with `outs`): per tensor an int32 [M] map (or None) — row m is written to row map[m] of its output (negative: dropped);
    the outputs then share one row stride and may have any number of rows (fvk_rmsnorm_rope_scatter_bf16)."""
    n = len(tensors)
    M, width = tensors[0].shape
    stride = tensors[0].stride(0)
    for t in tensors:
        _chk(t, BF16, "tensor")
        if t.shape != (M, width) or t.stride(1) != 1 or t.stride(0) != stride:
            raise RuntimeError("rmsnorm_rope: tensors must be [M,width] views with unit column stride and a common row stride")
    if outs is None:
        outs = [torch.empty((M, width), dtype=BF16, device=tensors[0].device) for _ in range(n)]
    elif len(outs) != n:
        raise RuntimeError("rmsnorm_rope: `outs` needs one output per tensor")
    ostride = outs[0].stride(0)
    for o in outs:  # caller-supplied outputs are written by raw pointer: same checks as the inputs
        _chk(o, BF16, "out")
        if o.dim() != 2 or o.shape[1] != width or o.stride(1) != 1 or o.stride(0) != ostride:
            raise RuntimeError("rmsnorm_rope: outputs must be [rows,width] bf16 views with unit column stride and a common row stride")
        if row_maps is None and o.shape[0] != M:
            raise RuntimeError(f"rmsnorm_rope: output has {o.shape[0]} rows, the inputs {M}")
    arr = C.c_void_p * n
    ins = arr(*[t.data_ptr() for t in tensors])
    os_ = arr(*[t.data_ptr() for t in outs])
    ws = None
    keep = []
    if weights is not None:
        keep = [None if w is None else _chk(w, BF16, "weight").contiguous() for w in weights]
        ws = arr(*[0 if w is None else w.data_ptr() for w in keep])
    cos, sin = _f32(cos, "cos"), _f32(sin, "sin")
    if row_maps is not None:
        if len(row_maps) != n or any(o.stride(0) != ostride or o.stride(1) != 1 for o in outs):
            raise RuntimeError("rmsnorm_rope: row_maps needs one entry per tensor and outputs with one common row stride")
        # contiguous maps only: a ``.contiguous()`` copy would be a fresh tensor object on every call, and _check_row_map (keyed on the
        # object) would pay its device sync per call instead of once per geometry
        if any(r is not None and not r.is_contiguous() for r in row_maps):
            raise RuntimeError("rmsnorm_rope: row maps must be contiguous int32 tensors")
        maps = [None if r is None else _chk(r, torch.int32, "row_map") for r in row_maps]
        if any(r is not None and r.numel() != M for r in maps):
            raise RuntimeError("rmsnorm_rope: every row map must have one entry per input row")
        for r, o in zip(maps, outs):
            if r is None:
                if o.shape[0] != M:
                    raise RuntimeError(f"rmsnorm_rope: an output without a row map needs {M} rows, got {o.shape[0]}")
            else:
                _check_row_map(r, o.shape[0])
        rm = arr(*[0 if r is None else r.data_ptr() for r in maps])
        _lib.call("fvk_rmsnorm_rope_scatter_bf16", ins, os_, ws, n, _p(cos), _p(sin), M, width, head_dim, seq_len or M, int(pos_offset), stride,
                  ostride, float(eps), rm, _stream())
        return outs
    _lib.call("fvk_rmsnorm_rope_bf16", ins, os_, ws, n, _p(cos), _p(sin), M, width, head_dim, seq_len or M, int(pos_offset), stride, ostride,
              float(eps), _stream())
    return outs


def qkv_norm_rope_pack(q, k, v, wq, wk, cos, sin, G, U, head_dim=128, seq_len=None, eps=1e-6, pos_offset=0, out=None, gate=None, heads_a=None):
    """Exchange #1 of the 2-D Ulysses sequence parallelism, packed by the norm / RoPE pass itself (fvk_qkv_norm_rope_pack_bf16).
    q, k, v: bf16 [Sl, width] views with unit column stride and one common row stride (column blocks of the fused QKV buffer).
    Returns the send buffer [G*U, Sl, 3, width // G]: per destination rank, per token, [K | V | Q] of that rank's head group.
    gate (the VSA compress gate, a fourth column block of the same buffer): [G*U, Sl, 4, width // G] = [K | V | Q | gate]
    (fvk_qkvg_norm_rope_pack_bf16).
    heads_a (dense path only): TWO send buffers — the first ``heads_a`` heads of every head group in [G*U, Sl, 3, heads_a*head_dim], the rest in
    [G*U, Sl, 3, (heads/G - heads_a)*head_dim] (fvk_qkv_norm_rope_pack2_bf16: the pipelined exchange's head chunks); returns the pair."""
    Sl, width = q.shape
    stride = q.stride(0)
    ns = 3 if gate is None else 4
    for t in (q, k, v) if gate is None else (q, k, v, gate):
        _chk(t, BF16, "tensor")
        if t.shape != (Sl, width) or t.stride(1) != 1 or t.stride(0) != stride:
            raise RuntimeError("qkv_norm_rope_pack: q, k, v must be [Sl,width] views with unit column stride and a common row stride")
    W = width // G
    if heads_a is not None:
        hg = W // head_dim
        if gate is not None or out is not None or not (0 < heads_a < hg):
            raise RuntimeError(f"qkv_norm_rope_pack: heads_a={heads_a} needs 0 < heads_a < {hg} heads per group, no gate and no `out`")
        wq_ = None if wq is None else _chk(wq, BF16, "wq").contiguous()
        wk_ = None if wk is None else _chk(wk, BF16, "wk").contiguous()
        sa = torch.empty((G * U, Sl, 3, heads_a * head_dim), dtype=BF16, device=q.device)
        sb = torch.empty((G * U, Sl, 3, (hg - heads_a) * head_dim), dtype=BF16, device=q.device)
        _lib.call("fvk_qkv_norm_rope_pack2_bf16", _p(q), _p(k), _p(v), _p(wq_), _p(wk_), _p(_f32(cos, "cos")), _p(_f32(sin, "sin")), _p(sa), _p(sb),
                  int(heads_a), Sl, width, head_dim, seq_len or Sl, int(pos_offset), stride, int(G), int(U), float(eps), _stream())
        return sa, sb
    if out is None:
        out = torch.empty((G * U, Sl, ns, W), dtype=BF16, device=q.device)
    elif tuple(out.shape) != (G * U, Sl, ns, W) or out.dtype != BF16 or not out.is_contiguous():
        raise RuntimeError(f"qkv_norm_rope_pack: `out` must be a contiguous bf16 [G*U, Sl, {ns}, width//G] buffer")
    wq = None if wq is None else _chk(wq, BF16, "wq").contiguous()
    wk = None if wk is None else _chk(wk, BF16, "wk").contiguous()
    cos, sin = _f32(cos, "cos"), _f32(sin, "sin")
    if gate is not None:
        _lib.call("fvk_qkvg_norm_rope_pack_bf16", _p(q), _p(k), _p(v), _p(gate), _p(wq), _p(wk), _p(cos), _p(sin), _p(out), Sl, width, head_dim,
                  seq_len or Sl, int(pos_offset), stride, int(G), int(U), float(eps), _stream())
        return out
    _lib.call("fvk_qkv_norm_rope_pack_bf16", _p(q), _p(k), _p(v), _p(wq), _p(wk), _p(cos), _p(sin), _p(out), Sl, width, head_dim,
              seq_len or Sl, int(pos_offset), stride, int(G), int(U), float(eps), _stream())
    return out


def v_transpose(v, src_rows=None):
    """v: bf16 [B,S,H,128] view (any strides with unit stride on D).  -> Vt [B,H,128,S_pad] (keys permuted, pad 0).
    src_rows (int32 [n]): key position p of Vt takes row src_rows[p] of v (negative = zero column), S_pad = round_up(n, 128) — the
    tile-major gather of V folded into the transpose (fvk_v_transpose_gather_bf16)."""
    _chk(v, BF16, "v")
    B, S, H, D = v.shape
    if v.stride(3) != 1:
        v = v.contiguous()
    n_keys = S if src_rows is None else src_rows.numel()
    S_pad = (n_keys + 127) // 128 * 128  # whole 128-key tiles: the 128-key-tile attention kernel reads V^T rows in 256-B pieces
    vt = torch.empty((B, H, D, S_pad), dtype=BF16, device=v.device)
    if src_rows is None:
        _lib.call("fvk_v_transpose_bf16", _p(v), _p(vt), B, S, H, D, v.stride(1), v.stride(0), v.stride(2), S_pad, _stream())
    else:
        src_rows = _chk(src_rows, torch.int32, "src_rows").contiguous()
        if src_rows.numel() != S_pad:
            raise RuntimeError(f"v_transpose: src_rows must cover whole 128-key tiles (got {src_rows.numel()} entries)")
        _lib.call("fvk_v_transpose_gather_bf16", _p(v), _p(vt), _p(src_rows), B, S, H, D, v.stride(1), v.stride(0), v.stride(2), S_pad, _stream())
    return vt


def gemm_vt_eligible(x, w, bias=None) -> bool:
    """Whether fvk_gemm_vt_bf16 serves V^T = w · x^T for token rows x [B, S, K] (or [S, K]) and V weight rows w [d, K].  Mirrors EVERY gate of
    the C side (gemm_bf16.hip: fvk_gemm_vt_bf16 -> gemm_pp_eligible / gemm_ph_eligible / gemm_w1_eligible), so that a caller which falls back
    to gemm + v_transpose on this predicate never meets FVK_ERR_ARG in the middle of a forward: 16-byte aligned operands (storage offsets of
    views count), B <= 65535, batch strides in whole 16-byte units, 32-bit byte offsets inside one 256-row panel."""
    K = x.shape[-1]
    S = x.shape[-2]
    B = x.shape[0] if x.dim() == 3 else 1
    d = w.shape[0]
    if not (x.dtype == BF16 and w.dtype == BF16 and x.dim() in (2, 3) and w.dim() == 2 and w.shape[1] == K):
        return False
    if not (K % 128 == 0 and K >= 128 and d % 128 == 0 and d > 128 and S > 0 and S % 8 == 0 and 1 <= B <= 65535):
        return False
    if not (x.is_contiguous() and w.is_contiguous()):
        return False
    if x.data_ptr() % 16 or w.data_ptr() % 16 or (bias is not None and bias.data_ptr() % 8):
        return False
    if (S * K) % 8:   # x batch stride in elements
        return False
    return 255 * K * 2 + K * 2 <= 0x7fffffff


def gemm_vt(x, w, bias=None):
    """V projection written straight into the attention kernels' V^T layout (fvk_gemm_vt_bf16): x bf16 [B, S, K] dense token rows, w bf16
    [d, K] = the V rows of the QKV weight, bias bf16 [d].  -> Vt [B, d / 128, 128, S_pad] — what ``v_transpose(linear(x, w, bias))`` returns,
    bit for bit, without writing V, reading it back and a layout pass."""
    _chk(x, BF16, "x"), _chk(w, BF16, "w")
    if x.dim() == 2:
        x = x[None]
    B, S, K = x.shape
    d = w.shape[0]
    if not gemm_vt_eligible(x, w, bias):
        raise RuntimeError(f"gemm_vt: shape not served (x {tuple(x.shape)}, w {tuple(w.shape)}): use gemm + v_transpose")
    if bias is not None:
        _chk(bias, BF16, "bias")
    S_pad = (S + 127) // 128 * 128
    vt = torch.empty((B, d // 128, 128, S_pad), dtype=BF16, device=x.device)
    _lib.call("fvk_gemm_vt_bf16", _p(w), _p(x), _p(bias), _p(vt), B, S, d, K, x.stride(1), x.stride(0), S_pad, _stream())
    return vt


# ------------------------------------------------------------------ GEMM
def gemm(x, w, bias=None, epilogue=EPI_NONE, residual=None, gate=None, rows_per_batch=None, out=None):
    """out[..., N] = epilogue(x[..., K] @ w[N, K]^T + bias).  x may be a 2-D view with a row stride."""
    _chk(x, BF16, "x"), _chk(w, BF16, "w")
    K = x.shape[-1]
    N = w.shape[0]
    if x.dim() == 2 and x.stride(1) == 1:
        x2 = x
    else:
        x2 = x.contiguous().view(-1, K)
    M = x2.shape[0]
    w = w.contiguous()
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=x.device)
    res2 = None
    if residual is not None:
        res2 = _chk(residual, BF16, "residual").contiguous().view(-1, N)
    gate = _f32(gate, "gate")
    if rows_per_batch is None:
        B = 1 if gate is None else gate.numel() // N
        rows_per_batch = max(M // B, 1)
    if bias is not None:
        bias = _chk(bias, BF16, "bias").contiguous()
    _lib.call("fvk_gemm_bf16", _p(x2), _p(w), _p(bias), _p(out), M, N, K, x2.stride(0), out.stride(0), epilogue, _p(res2),
              _p(gate), rows_per_batch, _stream())
    return out.view(*x.shape[:-1], N) if x.dim() != 2 else out


def gemm_batched(x, w, epilogue=EPI_NONE, scalar=1.0):
    """x [Bt, M, K], w [Bt, N, K] contiguous bf16 -> [Bt, M, N]."""
    _chk(x, BF16, "x"), _chk(w, BF16, "w")
    x, w = x.contiguous(), w.contiguous()
    Bt, M, K = x.shape
    N = w.shape[1]
    out = torch.empty((Bt, M, N), dtype=BF16, device=x.device)
    _lib.call("fvk_gemm_bf16_batched", _p(x), _p(w), _p(out), Bt, M, N, K, K, N, M * K, N * K, M * N, epilogue, float(scalar),
              _stream())
    return out


# ------------------------------------------------------------------ fp8 (e4m3fn) linear path
FP8 = torch.float8_e4m3fn


def fp8_quantize(x, rowwise=False):
    """Dynamic quantisation of a bf16 [M, K] tensor (fvk_fp8_quantize_bf16; ref: fp8_config.py:55-68).
    Returns (q float8_e4m3fn [M, K], scale fp32 [1] or [M, 1])."""
    _chk(x, BF16, "x")
    K = x.shape[-1]
    x2 = x if (x.dim() == 2 and x.stride(1) == 1) else x.contiguous().view(-1, K)
    M = x2.shape[0]
    q = torch.empty((M, K), dtype=FP8, device=x.device)
    n = M if rowwise else 1
    scale = torch.empty((n,), dtype=torch.float32, device=x.device)
    scratch = torch.empty((n,), dtype=torch.float32, device=x.device)
    _lib.call("fvk_fp8_quantize_bf16", _p(x2), _p(q), _p(scale), _p(scratch), M, K, x2.stride(0), int(rowwise), _stream())
    return q, (scale.view(M, 1) if rowwise else scale)


def gemm_fp8(x_q, x_scale, w_q, w_scale, bias=None, epilogue=EPI_NONE, residual=None, gate=None, rows_per_batch=None, out=None):
    """out = epilogue(bf16(bf16((x_q @ w_q^T) * x_scale * w_scale) + bias)) — torch._scaled_mm + bias of FP8QuantizeMethod.apply
    (fp8_config.py:119-158).  x_q [M,K], w_q [N,K] float8_e4m3fn; scales fp32 [1] (tensorwise) or [M,1] / [N] (rowwise)."""
    for t, n in ((x_q, "x_q"), (w_q, "w_q")):
        _chk(t, FP8, n)
    M, K = x_q.shape
    N = w_q.shape[0]
    x_q, w_q = x_q.contiguous(), w_q.contiguous()
    x_scale, w_scale = _f32(x_scale, "x_scale").view(-1), _f32(w_scale, "w_scale").view(-1)
    a_row, b_row = int(x_scale.numel() != 1), int(w_scale.numel() != 1)
    if (a_row and x_scale.numel() != M) or (b_row and w_scale.numel() != N):
        raise RuntimeError("gemm_fp8: scale shapes must be [1] or one per row / output channel")
    if out is None:
        out = torch.empty((M, N), dtype=BF16, device=x_q.device)
    res2 = None if residual is None else _chk(residual, BF16, "residual").contiguous().view(-1, N)
    gate = _f32(gate, "gate")
    if rows_per_batch is None:
        B = 1 if gate is None else gate.numel() // N
        rows_per_batch = max(M // B, 1)
    if bias is not None:
        bias = _chk(bias, BF16, "bias").contiguous()
    _lib.call("fvk_gemm_fp8", _p(x_q), _p(w_q), _p(x_scale), _p(w_scale), _p(bias), _p(out), M, N, K, out.stride(0), a_row, b_row,
              epilogue, _p(res2), _p(gate), rows_per_batch, _stream())
    return out


# ------------------------------------------------------------------ attention
def _attn_args(q, k, vt, o, scale, layout, lse=None, qk_dim=0):
    """layout 'bshd': q [B,S,H,D]; 'bhsd': q [B,H,S,D]."""
    for t, n in ((q, "q"), (k, "k"), (vt, "vt"), (o, "o")):
        _chk(t, BF16, n)
    if q.stride(-1) != 1 or k.stride(-1) != 1 or o.stride(-1) != 1:
        raise RuntimeError("attention: head_dim must be the unit-stride dimension")
    if layout == "bshd":
        B, Sq, H, D = q.shape
        Skv = k.shape[1]
        st = lambda t: (t.stride(0), t.stride(1), t.stride(2))
    else:
        B, H, Sq, D = q.shape
        Skv = k.shape[2]
        st = lambda t: (t.stride(0), t.stride(2), t.stride(1))
    if D != 128 and not (qk_dim == 384 and D == 384):
        raise RuntimeError(f"attention: head_dim {D} != 128")
    a = AttnArgs()
    a.q, a.k, a.vt, a.o = q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr()
    a.lse = None if lse is None else lse.data_ptr()
    a.B, a.H, a.Sq, a.Skv, a.Skv_pad = B, H, Sq, Skv, vt.shape[-1]
    a.q_bs, a.q_ss, a.q_hs = st(q)
    a.k_bs, a.k_ss, a.k_hs = st(k)
    a.o_bs, a.o_ss, a.o_hs = st(o)
    a.scale = float(scale)
    a.qk_dim = int(qk_dim)
    if qk_dim == 384:  # one 384-wide head = three 128-column output slices sharing q and k (include/fvk_amd.h)
        a.H, a.q_hs, a.k_hs, a.o_hs = 3 * H, 0, 0, 128
    return a


def _vt_of(v, layout):
    return v_transpose(v if layout == "bshd" else v.transpose(1, 2))


N_CUS = 256  # MI355X


def attn_key_splits(n_query_blocks: int, n_stages: int) -> int:
    """How many key runs fvk_attn_dense_split_bf16 should cut: 1 (no split) unless the 256-row workgroups leave more than a fifth of the
    last round of 256 CUs empty AND every run keeps at least 32 stages (4096 keys) — a partly filled chip is clock-compensated (the power
    the idle CUs do not draw goes into the busy ones' clock), so splitting pays only for clearly under-filled grids with long key axes:
    measured +9 % at 384 workgroups (SP = 4 on 12 heads -> 2 runs), +1.3 % at 192 (SP = 8 -> 4 runs), and a LOSS at 432 workgroups x 72
    stages (the 9x64x64 plumbing latent: 4 runs cost 12 % of the forward in merge traffic and per-run prologues)."""
    eff = lambda wg: wg / (N_CUS * -(-wg // N_CUS))
    best, best_eff = 1, eff(n_query_blocks)
    if best_eff >= 0.8:
        return 1
    for s in range(2, 9):
        if s * 32 > n_stages:
            break
        e = eff(n_query_blocks * s)
        if e > best_eff + 0.04:
            best, best_eff = s, e
    return best


def attn_key_splits_for(q, k, layout="bshd") -> int:
    """The key-run count attn_dense(key_splits=None) picks for these operands (1 = the plain kernels)."""
    if layout == "bshd":
        B, Sq, H, D = q.shape
        Skv = k.shape[1]
    else:
        B, H, Sq, D = q.shape
        Skv = k.shape[2]
    if Sq < 256 or D != 128:
        return 1
    return attn_key_splits(-(-Sq // 256) * H * B, -(-Skv // 128))


# fvk_attn_dense_kernel_bf16: the long-key kernel on 16x16x32 / 32x32x16 MFMAs (fixed softmax reference), or the 8-wave online-softmax kernel
# at any key length (the flash-attention rounding points; for parity-critical callers)
ATTN_KERNEL_DEFAULT, ATTN_KERNEL_W16, ATTN_KERNEL_W64, ATTN_KERNEL_PP2 = 0, 1, 2, 3


def attn_dense(q, k, v=None, vt=None, scale=None, layout="bshd", out=None, return_lse=False, key_splits=None, kernel=ATTN_KERNEL_DEFAULT):
    """Dense non-causal attention.  Pass v (same layout as k) or a precomputed vt = v_transpose(v).
    return_lse: also the base-2 log-sum-exp of the scaled scores per query row, fp32 [B, H, Sq] (full-length kernels only: Sq >= 256).
    key_splits: None = automatic (attn_key_splits: split-KV + merge for grids that do not fill the chip), 1 = never, n = that many runs.
    kernel: which long-key kernel (ATTN_KERNEL_*; same result to rounding — see include/fvk_amd.h for when either is faster)."""
    scale = q.shape[-1]**-0.5 if scale is None else scale
    if vt is None:
        vt = _vt_of(v, layout)
    o = torch.empty_like(q) if out is None else out
    lse = None
    if return_lse:
        B, H = (q.shape[0], q.shape[2]) if layout == "bshd" else (q.shape[0], q.shape[1])
        Sq = q.shape[1] if layout == "bshd" else q.shape[2]
        lse = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)
    a = _attn_args(q, k, vt, o, scale, layout, lse)
    if key_splits is None:
        key_splits = attn_key_splits_for(q, k, layout)
    if key_splits > 1:
        if kernel in (ATTN_KERNEL_W64, ATTN_KERNEL_PP2):
            # the split-KV form exists on attn_w16 only: never run a kernel other than the one the caller named (and may report)
            raise RuntimeError(f"attn_dense: kernel={kernel} cannot be combined with key_splits > 1 (the split-KV form runs attn_w16); "
                               "pass key_splits=1 or the default kernel")
        rows = a.B * a.H * a.Sq
        o_part = torch.empty((key_splits, rows, 128), dtype=torch.float32, device=q.device)
        lse_part = torch.empty((key_splits, rows), dtype=torch.float32, device=q.device)
        _lib.call("fvk_attn_dense_split_bf16", C.byref(a), int(key_splits), _p(o_part), _p(lse_part), _stream())
    elif kernel == ATTN_KERNEL_DEFAULT:
        _lib.call("fvk_attn_dense_bf16", C.byref(a), _stream())
    else:
        _lib.call("fvk_attn_dense_kernel_bf16", C.byref(a), int(kernel), _stream())
    return (o, lse) if return_lse else o


def attn_block_sparse(q, k, v, q2k_idx, q2k_num, kv_block_sizes, scale=None, layout="bhsd", return_lse=False, q_block=64, pair_union=False,
                      split_last_round=True, vt=None):
    """q2k_idx int32 [B,H,nq,max_kv] ascending block lists, q2k_num [B,H,nq], kv_block_sizes [nkv].
    pair_union (64-row lists, max_kv <= 2048): the two lists of neighbouring query blocks are merged (fvk_vsa_union_lists) and walked as one by
    their workgroup — a KV tile both selected is fetched once (fvk_attn_block_sparse_union_bf16); bit-identical output.  OFF by default: the
    walk fetches ~30 % fewer tiles on the selections the model makes but takes one step per tile of the UNION, both halves in lockstep, and the
    kernel is step-bound, not ingest-bound — 3.2 vs 2.65 ms per layer at cfg2 (profiles/r04u_vsa_union_ab.log).
    split_last_round (64-row lists): the workgroups of the launch's last, partly empty round walk their lists in 2-4 parts that are merged
    afterwards (fvk_attn_block_sparse_ws_bf16); False = every list whole.  vt: a ready V^T (v_transpose's layout; v is then ignored)."""
    scale = q.shape[-1]**-0.5 if scale is None else scale
    if vt is None:
        vt = _vt_of(v, layout)
    o = torch.empty_like(q)
    B, H = (q.shape[0], q.shape[2]) if layout == "bshd" else (q.shape[0], q.shape[1])
    Sq = q.shape[1] if layout == "bshd" else q.shape[2]
    lse = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device) if return_lse else None
    a = _attn_args(q, k, vt, o, scale, layout, lse)
    q2k_idx = _chk(q2k_idx, torch.int32, "q2k_idx").contiguous()
    q2k_num = _chk(q2k_num, torch.int32, "q2k_num").contiguous()
    kv_block_sizes = _chk(kv_block_sizes, torch.int32, "kv_block_sizes").contiguous()
    max_kv = q2k_idx.shape[-1]
    if pair_union and q_block == 64 and max_kv <= 2048:
        nq = q2k_idx.shape[-2]
        u_idx = torch.empty((B * H, (nq + 1) // 2, 2 * max_kv), dtype=torch.int32, device=q.device)
        u_num = torch.empty((B * H, (nq + 1) // 2), dtype=torch.int32, device=q.device)
        _lib.call("fvk_vsa_union_lists", _p(q2k_idx), _p(q2k_num), _p(kv_block_sizes), _p(u_idx), _p(u_num), B * H, nq, max_kv, _stream())
        _lib.call("fvk_attn_block_sparse_union_bf16", C.byref(a), _p(u_idx), _p(u_num), 2 * max_kv, _stream())
        return (o, lse) if return_lse else o
    # the split last round of the 64-row list kernel (include/fvk_amd.h: fvk_attn_block_sparse_ws_bf16) wants a workspace; one buffer per
    # (device, size), kept — a captured HIP graph replays with the address it was captured with
    ws_bytes = int(_lib.load().fvk_attn_block_sparse_workspace_bytes(C.byref(a), max_kv, int(q_block))) if split_last_round else 0
    ws = None
    if ws_bytes > 0:
        key = (q.device.index, ws_bytes)
        ws = _BS_WORKSPACES.get(key)
        if ws is None:
            ws = _BS_WORKSPACES[key] = torch.empty(ws_bytes, dtype=torch.uint8, device=q.device)
    _lib.call("fvk_attn_block_sparse_ws_bf16", C.byref(a), _p(q2k_idx), _p(q2k_num), _p(kv_block_sizes), max_kv, int(q_block),
              _p(ws) if ws is not None else None, ws_bytes, _stream())
    return (o, lse) if return_lse else o


_BS_WORKSPACES = {}


def attn_tile_lists(q, k, v, q2k_idx, q2k_num, kv_block_sizes, rows_per_list, q_rows_valid=None, scale=None, layout="bhsd", vt=None,
                    o_rows=None, n_out_rows=None):
    """Block-sparse attention with one KV block list per ``rows_per_list`` consecutive query rows (fvk_attn_tile_lists_bf16): sliding-tile
    windows in tile-major order.  q2k_idx int32 [B,H,Nl,max_kv], q2k_num int32 [B,H,Nl], kv_block_sizes int32 [Nkv],
    q_rows_valid int32 [Nl] or None.  vt: a ready V^T (then v is ignored).  o_rows int32 [Sq] + n_out_rows: query row r's output is stored at
    row o_rows[r] of an [.., n_out_rows, ..] output (negative: dropped; rows no entry points at are left unwritten)."""
    scale = q.shape[-1]**-0.5 if scale is None else scale
    if vt is None:
        vt = _vt_of(v, layout)
    if o_rows is None:
        o = torch.empty_like(q)
    else:
        o_rows = _chk(o_rows, torch.int32, "o_rows").contiguous()
        Sq = q.shape[1] if layout == "bshd" else q.shape[2]
        if o_rows.numel() != Sq or n_out_rows is None:
            raise RuntimeError("attn_tile_lists: o_rows needs one entry per query row and n_out_rows")
        shape = (q.shape[0], n_out_rows, q.shape[2], q.shape[3]) if layout == "bshd" else (q.shape[0], q.shape[1], n_out_rows, q.shape[3])
        o = torch.empty(shape, dtype=BF16, device=q.device)
    a = _attn_args(q, k, vt, o, scale, layout)
    q2k_idx = _chk(q2k_idx, torch.int32, "q2k_idx").contiguous()
    q2k_num = _chk(q2k_num, torch.int32, "q2k_num").contiguous()
    kv_block_sizes = _chk(kv_block_sizes, torch.int32, "kv_block_sizes").contiguous()
    if q_rows_valid is not None:
        q_rows_valid = _chk(q_rows_valid, torch.int32, "q_rows_valid").contiguous()
    _lib.call("fvk_attn_tile_lists_bf16", C.byref(a), _p(q2k_idx), _p(q2k_num), _p(kv_block_sizes), q2k_idx.shape[-1], int(rows_per_list),
              _p(q_rows_valid), _p(o_rows), _stream())
    return o


def attn_sta(q, k, v, canvas_tiles, tile_tokens, windows, scale=None, layout="bhsd"):
    """windows: list of (t,h,w) per head, in tiles."""
    scale = q.shape[-1]**-0.5 if scale is None else scale
    vt = _vt_of(v, layout)
    o = torch.empty_like(q)
    a = _attn_args(q, k, vt, o, scale, layout)
    flat = [int(x) for w in windows for x in w]
    arr = (C.c_int32 * len(flat))(*flat)
    _lib.call("fvk_attn_sta_bf16", C.byref(a), int(canvas_tiles[0]), int(canvas_tiles[1]), int(canvas_tiles[2]), int(tile_tokens),
              arr, _stream())
    return o


def attn_dense_wide(q, k, v, scale=None):
    """Single-head attention with head_dim 384 (the Wan VAE mid block, ref: wanvae.py:479-507).
    q, k, v: bf16 [S, 384] or [T, S, 384] views with unit column stride (e.g. the three column blocks of a fused [.., S, 1152] projection);
    a leading T = independent frames, ONE launch for all of them (a frame alone is 147 workgroups on 256 CUs)."""
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _chk(t, BF16, n)
    if q.dim() == 2:
        return attn_dense_wide(q[None], k[None], v[None], scale)[0]
    T, S, D = q.shape
    if D != 384 or k.shape != q.shape or v.shape != q.shape:
        raise RuntimeError(f"attn_dense_wide: q/k/v {tuple(q.shape)} {tuple(k.shape)} {tuple(v.shape)}: want three [T, S, 384] tensors")
    scale = D**-0.5 if scale is None else scale
    v4 = v.as_strided((T, S, 3, 128), (v.stride(0), v.stride(1), 128, 1))
    vt = v_transpose(v4)
    o = torch.empty((T, S, D), dtype=BF16, device=q.device)
    a = _attn_args(q.as_strided((T, S, 1, D), (q.stride(0), q.stride(1), 0, 1)), k.as_strided((T, S, 1, D), (k.stride(0), k.stride(1), 0, 1)), vt,
                   o.view(T, S, 1, D), scale, "bshd", qk_dim=384)
    _lib.call("fvk_attn_dense_bf16", C.byref(a), _stream())
    return o


# ------------------------------------------------------------------ Wan VAE decode
VAE_EPI_BIAS, VAE_EPI_RESIDUAL, VAE_EPI_FINAL = 0, 1, 2
# bench.py's roofline leg: set to a list to collect (taps "ktxksxks", Cin, Cout, algorithmic FLOPs, start event, end event) per conv
# launch, recorded on the launch stream around the kernel alone.  None (default) = no events, nothing extra on the stream.
VAE_CONV_EVENTS = None


def _conv_timed(kt, ks, Cin, Cout, T, H, W, launch):
    if VAE_CONV_EVENTS is None:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    VAE_CONV_EVENTS.append((f"{kt}x{ks}x{ks}", Cin, Cout, 2.0 * T * H * W * Cout * kt * ks * ks * Cin, e0, e1))


def vae_conv(inp, w, bias, *, T, H, W, kt, ks, ring_start=0, out=None, out_frame_stride=None, residual=None,
             res_frame_stride=None, upsample2x=False, out_f32=None, plane_stride=0):
    """Implicit-GEMM conv on channels-last frames (fvk_vae_conv_bf16).  inp: bf16 [ring, Hin, Win, Cin] contiguous;
    w: bf16 [Cout, kt*ks*ks*Cin]; out: bf16 tensor whose data_ptr is the first output pixel (frame stride in elements) or
    out_f32 for the final planar fp32 epilogue."""
    _chk(inp, BF16, "inp"), _chk(w, BF16, "w")
    ring, Hin, Win, Cin = inp.shape
    Cout = w.shape[0]
    if w.shape[1] != kt * ks * ks * Cin or not inp.is_contiguous() or not w.is_contiguous():
        raise RuntimeError(f"vae_conv: weight {tuple(w.shape)} does not match taps {kt}x{ks}x{ks} x Cin {Cin} (or non-contiguous input)")
    if (Hin, Win) != ((H // 2, W // 2) if upsample2x else (H, W)):
        raise RuntimeError(f"vae_conv: input spatial size {(Hin, Win)} inconsistent with output {(H, W)}")
    if out_f32 is not None:
        epi = VAE_EPI_FINAL
        _chk(out_f32, torch.float32, "out_f32")
    else:
        epi = VAE_EPI_RESIDUAL if residual is not None else VAE_EPI_BIAS
        if out is None:
            out = torch.empty((T, H, W, Cout), dtype=BF16, device=inp.device)
        if out_frame_stride is None:
            out_frame_stride = H * W * Cout
        if residual is not None:
            _chk(residual, BF16, "residual")
            if res_frame_stride is None:
                res_frame_stride = H * W * Cout
    if bias is not None:
        _chk(bias, BF16, "bias")
    _conv_timed(kt, ks, Cin, Cout, T, H, W, lambda: _lib.call(
        "fvk_vae_conv_bf16", _p(inp), _p(w), _p(bias), _p(out), _p(residual), _p(out_f32), T, H, W, Cin, Cout, kt, ks, ks,
        ring, int(ring_start), int(out_frame_stride or 0), int(res_frame_stride or 0), int(plane_stride), int(upsample2x), epi,
        _stream()))
    return out if out_f32 is None else out_f32


def vae_conv_norm(inp, w, bias, gamma, norm_ring_buf, *, T, H, W, kt, norm_slot0, ring_start=0, out=None, out_frame_stride=None, residual=None,
                  res_frame_stride=None, upsample2x=False, silu=True, want_raw=True):
    """3x3-tap conv (Cout = 96 or 192) whose epilogue also writes WanRMS_norm(+SiLU) of the result into the consumer conv's input ring
    (fvk_vae_conv_norm_bf16).  Returns the un-normed output [T,H,W,Cout] (None with want_raw=False: nothing but the ring is written)."""
    _chk(inp, BF16, "inp"), _chk(w, BF16, "w"), _chk(norm_ring_buf, BF16, "norm_ring_buf")
    gamma = _f32(gamma, "gamma")
    ring, Hin, Win, Cin = inp.shape
    Cout = w.shape[0]
    if w.shape[1] != kt * 9 * Cin or not inp.is_contiguous() or not w.is_contiguous() or not norm_ring_buf.is_contiguous():
        raise RuntimeError(f"vae_conv_norm: weight {tuple(w.shape)} does not match taps {kt}x3x3 x Cin {Cin} (or non-contiguous tensors)")
    if (Hin, Win) != ((H // 2, W // 2) if upsample2x else (H, W)):
        raise RuntimeError(f"vae_conv_norm: input spatial size {(Hin, Win)} inconsistent with output {(H, W)}")
    if tuple(norm_ring_buf.shape[1:]) != (H, W, Cout):
        raise RuntimeError(f"vae_conv_norm: consumer ring {tuple(norm_ring_buf.shape)} does not hold [*, {H}, {W}, {Cout}] frames")
    if want_raw:
        if out is None:
            out = torch.empty((T, H, W, Cout), dtype=BF16, device=inp.device)
        if out_frame_stride is None:
            out_frame_stride = H * W * Cout
    else:
        out = None
    if residual is not None:
        _chk(residual, BF16, "residual")
        if res_frame_stride is None:
            res_frame_stride = H * W * Cout
    if bias is not None:
        _chk(bias, BF16, "bias")
    _conv_timed(kt, 3, Cin, Cout, T, H, W, lambda: _lib.call(
        "fvk_vae_conv_norm_bf16", _p(inp), _p(w), _p(bias), _p(out), _p(residual), T, H, W, Cin, Cout, kt, ring, int(ring_start),
        int(out_frame_stride or 0), int(res_frame_stride or 0), int(upsample2x), _p(gamma), _p(norm_ring_buf), norm_ring_buf.shape[0],
        int(norm_slot0), int(silu), _stream()))
    return out


def vae_blend(a, b, extent, axis):
    """In-place linear cross-fade of tile b's leading `extent` slices along `axis` with tile a's trailing ones
    (fvk_vae_blend_f32; the reference's blend_t / blend_v / blend_h, common.py:94-114).  a, b: fp32 [C, T, H, W] tiles (or views
    of them sliced along T): dims after `axis` contiguous, dims 1..axis-1 collapsing to one stride."""
    _chk(a, torch.float32, "a"), _chk(b, torch.float32, "b")
    if a.dim() != 4 or b.dim() != 4 or axis not in (1, 2, 3):
        raise RuntimeError("vae_blend: expects [C,T,H,W] tiles and axis 1 (T), 2 (H) or 3 (W)")
    if any(a.shape[d] != b.shape[d] for d in range(4) if d != axis):
        raise RuntimeError(f"vae_blend: tiles {tuple(a.shape)} / {tuple(b.shape)} differ off the blend axis {axis}")
    if a.numel() == 0 or b.numel() == 0 or extent <= 0:
        return b  # min(len_a, len_b, extent) == 0: the reference's loop body never runs (common.py:95-96)
    inner = 1
    for d in range(3, axis, -1):
        if a.stride(d) != inner or b.stride(d) != inner:
            raise RuntimeError("vae_blend: dims after the blend axis must be contiguous")
        inner *= a.shape[d]
    outer1 = 1
    for d in range(1, axis):
        outer1 *= a.shape[d]

    def stride1(t):
        for d in range(1, axis - 1):
            if t.stride(d) != t.stride(d + 1) * t.shape[d + 1]:
                raise RuntimeError("vae_blend: dims between the channel and the blend axis do not collapse to one stride")
        return t.stride(axis - 1) if axis > 1 else 0

    _lib.call("fvk_vae_blend_f32", _p(a), _p(b), a.shape[0], outer1, inner, a.shape[axis], b.shape[axis], int(extent), a.stride(0),
              stride1(a), a.stride(axis), b.stride(0), stride1(b), b.stride(axis), _stream())
    return b


def vae_postprocess_u8(pixels):
    """fp32 planar pixels [3, T, H, W] (or [1, 3, T, H, W]) in [-1, 1] -> uint8 frames [T, H, W, 3]
    (fvk_vae_postprocess_u8: (x/2 + 0.5).clamp(0,1) * 255 -> clamp -> truncate; decoding.py:210, video_generator.py:912-913)."""
    _chk(pixels, torch.float32, "pixels")
    if pixels.dim() == 5:
        if pixels.shape[0] != 1:
            raise RuntimeError("vae_postprocess_u8: batch must be 1")
        pixels = pixels[0]
    Cc, T, H, W = pixels.shape
    if Cc != 3 or not pixels[0].is_contiguous() or (pixels.stride(0) < T * H * W):
        raise RuntimeError(f"vae_postprocess_u8: expected planar [3,T,H,W] with contiguous planes, got {tuple(pixels.shape)} strides {pixels.stride()}")
    out = torch.empty((T, H, W, 3), dtype=torch.uint8, device=pixels.device)
    _lib.call("fvk_vae_postprocess_u8", _p(pixels), _p(out), T, H, W, pixels.stride(0), _stream())
    return out


def vae_rmsnorm_silu(x, gamma, out_ring, *, HW, slot0=0, silu=True):
    """x bf16 [n_pix, C] (contiguous) -> out_ring [ring, HW, C] at frame slots (slot0 + t) % ring (fvk_vae_rmsnorm_silu_bf16)."""
    _chk(x, BF16, "x"), _chk(out_ring, BF16, "out_ring")
    gamma = _f32(gamma, "gamma")
    C_ = x.shape[-1]
    n_pix = x.numel() // C_
    ring = out_ring.numel() // (HW * C_)
    if not x.is_contiguous() or not out_ring.is_contiguous():
        raise RuntimeError("vae_rmsnorm_silu: tensors must be contiguous")
    _lib.call("fvk_vae_rmsnorm_silu_bf16", _p(x), _p(gamma), _p(out_ring), n_pix, C_, HW, ring, int(slot0), int(silu), _stream())
    return out_ring


# ------------------------------------------------------------------ VSA pieces
def vsa_build_metadata_host(dit_seq_shape, tile_size=(4, 4, 4)):
    """Pure-integer metadata through the C ABI (CPU).  Returns dict of int32 torch CPU tensors."""
    T, H, W = dit_seq_shape
    n = T * H * W
    nt = [math.ceil(s / t) for s, t in zip(dit_seq_shape, tile_size)]
    perm, rev, npi, unt = (torch.empty(n, dtype=torch.int32) for _ in range(4))
    vbs = torch.empty(nt[0] * nt[1] * nt[2], dtype=torch.int32)
    _lib.call("fvk_vsa_build_metadata_host", T, H, W, *tile_size, _p(perm), _p(rev), _p(vbs), _p(npi), _p(unt))
    return dict(tile_partition_indices=perm, reverse_tile_partition_indices=rev, variable_block_sizes=vbs,
                non_pad_index=npi, untile_combined_index=unt, num_tiles=tuple(nt))


def gather_rows(src, n_dst_rows, src_index=None, dst_index=None, zero_init=False, out=None):
    """src bf16 [B, Ns, ...row...] -> dst [B, n_dst_rows, ...row...]; dst[:, dst_index[i]] = src[:, src_index[i]].
    src may be a view whose rows ([...row...] contiguous) are a fixed stride apart (a column block of the fused QKV buffer); ``out`` = a
    preallocated contiguous destination (e.g. a persistent tile buffer whose pad rows were zeroed once)."""
    _chk(src, BF16, "src")
    B = src.shape[0]
    row = src[0, 0].numel()
    inner_ok = src[0, 0].is_contiguous() and (B == 1 or src.stride(0) % 8 == 0) and src.stride(1) % 8 == 0 and src.stride(1) >= row
    if not inner_ok:
        src = src.contiguous()
    if out is None:
        alloc = torch.zeros if zero_init else torch.empty
        out = alloc((B, n_dst_rows, *src.shape[2:]), dtype=BF16, device=src.device)
    elif out.shape != (B, n_dst_rows, *src.shape[2:]) or out.dtype != BF16 or not out.is_contiguous():
        raise RuntimeError("gather_rows: `out` must be a contiguous bf16 tensor of the destination shape")
    any_index = src_index if src_index is not None else dst_index
    n = n_dst_rows if any_index is None else any_index.numel()
    si = None if src_index is None else _chk(src_index, torch.int32, "src_index")
    di = None if dst_index is None else _chk(dst_index, torch.int32, "dst_index")
    _lib.call("fvk_gather_rows_strided_bf16", _p(src), _p(out), _p(si), _p(di), B, n, row, src.stride(1), row, src.stride(0), out.stride(0),
              _stream())
    return out


def block_mean(x, vbs, block=64, layout="bhsd", src_rows=None):
    """x [B,H,S_pad,D] (or bshd) -> [B,H,Nblk,D].  src_rows (int32 [Nblk * block]): tile-major row p is row src_rows[p] of x (negative: a
    padding row) — x is then in TOKEN order and the tile gather is folded in (fvk_block_mean_gather_bf16, bit-identical)."""
    _chk(x, BF16, "x")
    if x.stride(-1) != 1:
        raise RuntimeError("block_mean: head_dim must be the unit-stride dimension")
    if layout == "bhsd":
        B, H, S, D = x.shape
        bs, hs, ss = x.stride(0), x.stride(1), x.stride(2)
    else:
        B, S, H, D = x.shape
        bs, ss, hs = x.stride(0), x.stride(1), x.stride(2)
    vbs = _chk(vbs, torch.int32, "vbs").contiguous()
    if src_rows is not None:
        src_rows = _chk(src_rows, torch.int32, "src_rows").contiguous()
        nb = vbs.numel()
        if src_rows.numel() < nb * block:
            raise RuntimeError(f"block_mean: src_rows holds {src_rows.numel()} entries, {nb} blocks of {block} rows need {nb * block}")
        out = torch.empty((B, H, nb, D), dtype=BF16, device=x.device)
        _lib.call("fvk_block_mean_gather_bf16", _p(x), _p(out), _p(vbs), _p(src_rows), B, H, nb, block, D, bs, ss, hs, _stream())
        return out
    nb = S // block
    out = torch.empty((B, H, nb, D), dtype=BF16, device=x.device)
    _lib.call("fvk_block_mean_bf16", _p(x), _p(out), _p(vbs), B, H, nb, block, D, bs, ss, hs, _stream())
    return out


def topk_mask(scores, topk):
    if scores.dtype not in (BF16, torch.float32) or not scores.is_cuda:
        raise RuntimeError("topk_mask: scores must be a bf16/fp32 device tensor")
    s = scores.contiguous()
    n = s.shape[-1]
    mask = torch.empty(s.shape, dtype=torch.uint8, device=s.device)
    _lib.call("fvk_topk_mask", _p(s), int(s.dtype == torch.float32), _p(mask), s.numel() // n, n, int(topk), _stream())
    return mask.view(torch.bool)


def map_to_index(mask):
    m = mask.contiguous().view(torch.uint8)
    n = m.shape[-1]
    idx = torch.empty(m.shape, dtype=torch.int32, device=m.device)
    num = torch.empty(m.shape[:-1], dtype=torch.int32, device=m.device)
    _lib.call("fvk_map_to_index", _p(m), _p(idx), _p(num), m.numel() // n, n, _stream())
    return idx, num


def softmax_rows(x):
    _chk(x, BF16, "x")
    x = x.contiguous()
    out = torch.empty_like(x)
    _lib.call("fvk_softmax_rows_bf16", _p(x), _p(out), x.numel() // x.shape[-1], x.shape[-1], _stream())
    return out


def vsa_combine(out_c, out_s, gate=None, block=64, layout="bhsd", token_of_row=None, n_tokens=None):
    """out_c [B,H,Nblk,D] contiguous; out_s / gate [B,H,S,D] (bhsd) or [B,S,H,D] (bshd).
    token_of_row (int32 [S]) + n_tokens: ``gate`` is in TOKEN order ([B, n_tokens, H, D] / [B, H, n_tokens, D], any strides with unit D
    stride) and the result comes back in token order — the tile gather of the gate and the untile gather of the output folded in
    (fvk_vsa_combine_scatter_bf16); rows with a negative entry are padding."""
    _chk(out_c, BF16, "out_c"), _chk(out_s, BF16, "out_s")
    out_c = out_c.contiguous()
    out_s = out_s.contiguous()
    if gate is not None and token_of_row is None:
        gate = _chk(gate, BF16, "gate").contiguous()
    if layout == "bhsd":
        B, H, S, D = out_s.shape
        bs, hs, ss = out_s.stride(0), out_s.stride(1), out_s.stride(2)
    else:
        B, S, H, D = out_s.shape
        bs, ss, hs = out_s.stride(0), out_s.stride(1), out_s.stride(2)
    if token_of_row is not None:
        token_of_row = _chk(token_of_row, torch.int32, "token_of_row").contiguous()
        if token_of_row.numel() != S or n_tokens is None:
            raise RuntimeError("vsa_combine: token_of_row needs one entry per tile-major row and n_tokens")
        out = torch.empty((B, H, n_tokens, D) if layout == "bhsd" else (B, n_tokens, H, D), dtype=BF16, device=out_s.device)
        st = (lambda t: (t.stride(0), t.stride(2), t.stride(1))) if layout == "bhsd" else (lambda t: (t.stride(0), t.stride(1), t.stride(2)))
        if gate is not None:
            _chk(gate, BF16, "gate")
            if gate.stride(-1) != 1 or gate.shape != out.shape:
                raise RuntimeError("vsa_combine: token-order gate must match the output shape with a unit head_dim stride")
        g_st = st(gate) if gate is not None else (0, 0, 0)
        _lib.call("fvk_vsa_combine_scatter_bf16", _p(out_c), _p(out_s), _p(gate), _p(out), _p(token_of_row), B, S, H, D, block, bs, ss, hs,
                  *g_st, *st(out), _stream())
        return out
    out = torch.empty_like(out_s)
    _lib.call("fvk_vsa_combine_bf16", _p(out_c), _p(out_s), _p(gate), _p(out), B, S, H, D, block, bs, ss, hs, _stream())
    return out


# ------------------------------------------------------------------ glue
def patchify(latent, patch=(1, 2, 2)):
    _chk(latent, BF16, "latent")
    latent = latent.contiguous()
    B, Cc, T, Hh, W = latent.shape
    pt, ph, pw = patch
    out = torch.empty((B, (T // pt) * (Hh // ph) * (W // pw), Cc * pt * ph * pw), dtype=BF16, device=latent.device)
    _lib.call("fvk_patchify_bf16", _p(latent), _p(out), B, Cc, T, Hh, W, pt, ph, pw, _stream())
    return out


def unpatchify(x, latent_shape, patch=(1, 2, 2)):
    _chk(x, BF16, "x")
    x = x.contiguous()
    B, Cc, T, Hh, W = latent_shape
    out = torch.empty(latent_shape, dtype=BF16, device=x.device)
    _lib.call("fvk_unpatchify_bf16", _p(x), _p(out), B, Cc, T, Hh, W, *patch, _stream())
    return out


def timestep_embedding(t, dim, max_period=10000.0):
    t = t.to(device=t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device()), dtype=torch.float32).contiguous()
    _chk(t, torch.float32, "t")
    out = torch.empty((t.shape[0], dim), dtype=BF16, device=t.device)
    _lib.call("fvk_timestep_embedding_bf16", _p(t), _p(out), t.shape[0], dim, float(max_period), _stream())
    return out


def silu(x):
    _chk(x, BF16, "x")
    x = x.contiguous()
    out = torch.empty_like(x)
    _lib.call("fvk_silu_bf16", _p(x), _p(out), x.numel(), _stream())
    return out


def mfma_sustained_probe(seconds: float = 1.0, data: int = 1, workgroups: int = 256, iters: int = 20000):
    """Measurement (bench.py): the registers-only 16x16x32 bf16 MFMA stream of fvk_mfma_sustained_probe_bf16, launched back to back for about
    ``seconds`` on the current stream.  Returns (TFLOP/s, launches, ms per launch)."""
    import time
    dev = torch.device("cuda", torch.cuda.current_device())
    out = torch.empty(workgroups * 256, dtype=torch.float32, device=dev)
    fl = workgroups * 4.0 * iters * 64 * 16384
    for _ in range(3):  # warm-up: clock ramp and module load
        _lib.call("fvk_mfma_sustained_probe_bf16", _p(out), workgroups, iters, data, _stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n, t0 = 0, time.perf_counter()
    e0.record()
    while True:
        for _ in range(8):
            _lib.call("fvk_mfma_sustained_probe_bf16", _p(out), workgroups, iters, data, _stream())
        n += 8
        torch.cuda.synchronize()
        if time.perf_counter() - t0 >= seconds:
            break
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    return fl / (ms * 1e-3) / 1e12, n, ms

