"""The C-ABI library loads on a GPU-less host and exports every symbol include/fvk_amd.h declares; the pure-CPU
entry point (VSA metadata) is bit-exact against the oracle and the golden fixtures.  No compute kernels are launched."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import vsa_oracle as V

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as G
    G.build()
    from fastvideo_amd import _lib
    return _lib


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "fvk_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(fvk_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    cdll = lib.load()
    for name in sorted(declared):
        assert hasattr(cdll, name), f"{name} declared in include/fvk_amd.h but not exported"
    assert declared == set(lib.SIGNATURES), "ctypes SIGNATURES table out of sync with the header"
    assert cdll.fvk_abi_version() == lib.ABI_VERSION


def test_errors_surface_as_runtime_error(lib):
    with pytest.raises(RuntimeError, match="bad shape"):
        lib.call("fvk_vsa_build_metadata_host", 0, 1, 1, 4, 4, 4, None, None, None, None, None)


def test_argument_checks_of_the_newer_entries(lib):
    """Argument validation happens on the host before any launch, so it is testable without a GPU: every refusal names its entry point."""
    import ctypes as C
    buf = (C.c_char * 4096)()
    p = C.cast(buf, C.c_void_p)
    with pytest.raises(RuntimeError, match="fvk_dmd_step"):      # noise / sigma_next / next_out must come together
        lib.call("fvk_dmd_step", p, p, 0, p, p, None, p, None, 1, 8, None)
    with pytest.raises(RuntimeError, match="fvk_dmd_step"):
        lib.call("fvk_dmd_step", None, p, 0, p, None, None, p, None, 1, 8, None)
    with pytest.raises(RuntimeError, match="fvk_gather_rows_strided_bf16"):  # row stride smaller than the row
        lib.call("fvk_gather_rows_strided_bf16", p, p, None, None, 1, 4, 64, 32, 64, 0, 0, None)
    with pytest.raises(RuntimeError, match="96- and 192-channel"):          # the fused norm needs all channels of a pixel in one workgroup
        lib.call("fvk_vae_conv_norm_bf16", p, p, None, p, None, 1, 8, 8, 32, 384, 3, 3, 0, 0, 0, 0, p, p, 3, 0, 1, None)
    with pytest.raises(RuntimeError, match="fvk_vae_conv_norm_bf16"):       # consumer ring shorter than the chunk
        lib.call("fvk_vae_conv_norm_bf16", p, p, None, p, None, 4, 8, 8, 32, 96, 3, 6, 0, 0, 0, 0, p, p, 3, 0, 1, None)
    # round-2 entries: scattered / gathered row maps and the shared-list attention
    arr = (C.c_void_p * 1)(p.value)
    with pytest.raises(RuntimeError, match="fvk_rmsnorm_rope_scatter_bf16"):   # the scatter form needs its row-map array
        lib.call("fvk_rmsnorm_rope_scatter_bf16", arr, arr, None, 1, None, None, 4, 128, 128, 4, 0, 128, 128, 1e-6, None, None)
    with pytest.raises(RuntimeError, match="fvk_v_transpose_gather_bf16"):
        lib.call("fvk_v_transpose_gather_bf16", p, p, None, 1, 64, 1, 128, 128, 0, 128, 64, None)
    with pytest.raises(RuntimeError, match="fvk_vsa_combine_scatter_bf16"):
        lib.call("fvk_vsa_combine_scatter_bf16", p, p, None, p, None, 1, 64, 1, 128, 64, 0, 128, 128, 0, 128, 128, 0, 128, 128, None)
    with pytest.raises(RuntimeError, match="strides must keep 16-byte alignment"):
        lib.call("fvk_vsa_combine_scatter_bf16", p, p, None, p, p, 1, 64, 1, 128, 64, 0, 128, 128, 0, 130, 128, 0, 128, 128, None)
    # round-3 entries: split-KV attention (workspace and run count checked before any launch), 4-slot exchange packing
    from fastvideo_amd._lib import AttnArgs
    aa = AttnArgs()
    aa.q = aa.k = aa.vt = aa.o = p.value
    aa.B, aa.H, aa.Sq, aa.Skv, aa.Skv_pad = 1, 1, 256, 256, 256
    aa.q_ss = aa.k_ss = aa.o_ss = 128
    aa.scale = 0.1
    with pytest.raises(RuntimeError, match="n_split"):
        lib.call("fvk_attn_dense_split_bf16", C.byref(aa), 1, p, p, None)
    with pytest.raises(RuntimeError, match="null workspace"):
        lib.call("fvk_attn_dense_split_bf16", C.byref(aa), 4, None, p, None)
    with pytest.raises(RuntimeError, match="kernel=7"):   # the long-key kernel is 0 (default), 1 (attn_w16) or 2 (attn_w64)
        lib.call("fvk_attn_dense_kernel_bf16", C.byref(aa), 7, None)
    with pytest.raises(RuntimeError, match="null gate"):
        lib.call("fvk_qkvg_norm_rope_pack_bf16", p, p, p, None, None, None, None, None, p, 4, 128, 128, 4, 0, 128, 1, 1, 1e-6, None)
    with pytest.raises(RuntimeError, match="unknown tunable"):
        lib.call("fvk_set_tunable", b"no_such_knob", 1)
    lib.call("fvk_set_tunable", b"vsa_impl", 0)
    with pytest.raises(RuntimeError, match="libfvk_probe"):   # the product library holds the shipped configuration only
        lib.call("fvk_set_tunable", b"vsa_impl", 1)
    assert lib.load().fvk_is_probe_build() == 0


def test_tile_lists_attention_argument_checks(lib):
    """fvk_attn_tile_lists_bf16 validates its geometry on the host (no launch): rows_per_list a multiple of 128 and >= 256, Sq a multiple of
    it, scattered output rows only with whole 256-row groups."""
    import ctypes as C
    buf = (C.c_char * 4096)()
    p = C.cast(buf, C.c_void_p)
    a = lib.AttnArgs()
    a.q = a.k = a.vt = a.o = p.value
    a.B, a.H, a.Sq, a.Skv, a.Skv_pad = 1, 1, 768, 256, 256
    a.q_ss = a.k_ss = a.o_ss = 128
    a.q_hs = a.k_hs = a.o_hs = 128
    a.scale = 1.0
    for rows, msg in ((128, "rows_per_list=128"), (320, "rows_per_list=320"), (512, "multiple of rows_per_list=512")):
        with pytest.raises(RuntimeError, match=msg):
            lib.call("fvk_attn_tile_lists_bf16", C.byref(a), p, p, p, 4, rows, None, None, None)
    with pytest.raises(RuntimeError, match="o_rows"):
        lib.call("fvk_attn_tile_lists_bf16", C.byref(a), p, p, p, 4, 384, None, p, None)
    with pytest.raises(RuntimeError, match="null index arrays"):
        lib.call("fvk_attn_tile_lists_bf16", C.byref(a), None, p, p, 4, 384, None, None, None)


def test_attention_refuses_slices_beyond_the_32bit_descriptor_range(lib):
    """The K / V^T streams are addressed with 32-bit buffer descriptors and byte offsets per (batch, head) slice (attn_pp2.hip,
    attn_fwd.hip).  Wan2.2-A14B at 129f x 720p with K read in place from a fused [S, 4*5120] QKV+gate buffer is 138 240 keys x
    40 960 B = 5.7 GB: offsets would wrap silently.  The host check refuses it before any launch (so it is testable here)."""
    import ctypes as C
    buf = (C.c_char * 4096)()
    p = C.cast(buf, C.c_void_p).value

    def args(skv, k_ss, h=40):
        a = lib.AttnArgs()
        a.q = a.k = a.vt = a.o = p
        a.lse = None
        a.B, a.H, a.Sq, a.Skv, a.Skv_pad = 1, h, skv, skv, (skv + 127) // 128 * 128
        a.q_bs = a.k_bs = a.o_bs = 0
        a.q_ss, a.q_hs, a.k_ss, a.k_hs, a.o_ss, a.o_hs = h * 128, 128, k_ss, 128, h * 128, 128
        a.scale, a.qk_dim = 128**-0.5, 0
        return a

    bad = args(138240, 4 * 5120)
    for entry, extra in (("fvk_attn_dense_bf16", ()), ("fvk_attn_block_sparse_bf16", (p, p, p, 8, 64)),
                         ("fvk_attn_sta_bf16", (6, 6, 10, 384, (C.c_int32 * 120)(*([3] * 120))))):
        with pytest.raises(RuntimeError, match="4 GiB"):
            lib.call(entry, C.byref(bad), *extra, None)
    # the same keys in a contiguous [S, H, D] copy (what wan_dit.py hands over: 10 240 B per token) are inside the range:
    # the check passes and the call proceeds to the launch, which is what fails on a GPU-less host
    if not torch.cuda.is_available():  # (with a device present the call would really launch on the dummy pointers)
        ok = args(138240, 5120)
        try:
            lib.call("fvk_attn_dense_bf16", C.byref(ok), None)
        except RuntimeError as e:
            assert "4 GiB" not in str(e)


@pytest.mark.parametrize("shape", [(8, 16, 16), (9, 10, 7), (5, 7, 3), (2, 2, 2), (21, 30, 52)])
def test_vsa_metadata_host_bit_exact(lib, shape):
    from fastvideo_amd import ops
    m = ops.vsa_build_metadata_host(shape)
    ref = V.build_metadata(tuple(s * p for s, p in zip(shape, (1, 2, 2))))
    for key in ("tile_partition_indices", "reverse_tile_partition_indices", "variable_block_sizes", "non_pad_index",
                "untile_combined_index"):
        assert np.array_equal(m[key].numpy(), ref[key]), key
    assert m["num_tiles"] == ref["num_tiles"]


def test_vsa_metadata_host_golden(lib, golden_dir):
    from fastvideo_amd import ops
    z = np.load(os.path.join(golden_dir, "vsa_meta.npz"))
    for lat in sorted({k.split("/")[0] for k in z.files}):
        t, h, w = (int(v) for v in lat.split("x"))
        m = ops.vsa_build_metadata_host((t, h // 2, w // 2))
        assert np.array_equal(m["tile_partition_indices"].numpy(), z[lat + "/perm"])
        assert np.array_equal(m["untile_combined_index"].numpy(), z[lat + "/untile"])
        assert np.array_equal(m["variable_block_sizes"].numpy(), z[lat + "/vbs"])


def test_kernel_api_index_helpers(lib):
    from fastvideo_amd import kernel_api as KA
    assert KA.get_tile_partition_indices((2, 2, 2), (2, 2, 2)).tolist() == list(range(8))
    md = KA.build_vsa_metadata((9, 10, 7))
    assert md["num_tiles"] == (3, 3, 2) and md["max_block_size"] == 64
    with pytest.raises(ValueError, match="Unsupported VSA tile volume 27"):
        KA.build_vsa_metadata((6, 6, 6), tile_size=(3, 3, 3))
    idx = KA.get_non_pad_index(torch.tensor([20, 40]), 64)
    assert idx[0].item() == 0 and idx[20].item() == 64
    f, r = KA.get_tile_partition_indices((9, 10, 7)), KA.get_reverse_tile_partition_indices((9, 10, 7))
    assert torch.equal(r[f], torch.arange(630))


def test_no_cpu_fallback(lib):
    from fastvideo_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(4, 64, dtype=torch.bfloat16))


def test_measurement_build_loads_and_exports_the_same_abi():
    """scripts/probes/libfvk_probe.so (built by build()): the product sources + the non-shipping variants behind fvk_set_tunable.  It
    must export every symbol of include/fvk_amd.h too (the measurement scripts run the whole Python binding on it) and say what it is."""
    import subprocess
    import sys
    code = ("import os; os.environ['FVK_PROBE_LIB']='1'\n"
            "from fastvideo_amd import _lib\n"
            "l=_lib.load(); assert l.fvk_is_probe_build()==1 and _lib.LIB_PATH.endswith('libfvk_probe.so')\n"
            "_lib.call('fvk_set_tunable', b'attn_impl', 103); _lib.call('fvk_set_tunable', b'attn_impl', 0); print('probe-ok')")
    r = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "probe-ok" in r.stdout, r.stdout[-2000:]
