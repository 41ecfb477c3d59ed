"""ctypes binding of libfvk_amd.so (the C ABI declared in include/fvk_amd.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised
(the reference's convention for a refused backend is ValueError/ImportError at selection time and
TORCH_CHECK -> RuntimeError at call time; fastvideo/platforms/cuda.py:149-154,
fastvideo-kernel/csrc/attention/st_attn_h100.cu:386-411)."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# FVK_PROBE_LIB=1 (measurement scripts and the variant tests only): bind the measurement build of the same sources, which also holds
# the non-shipping kernel variants behind fvk_set_tunable (scripts/probes/libfvk_probe.so, built by _build.build_probe()).
_PROBE_SEL = os.environ.get("FVK_PROBE_LIB", "")
PROBE = _PROBE_SEL == "1" or _PROBE_SEL.startswith("bug")
# "bug", "bug2", "bug_s<N>" (scripts/coresidency_*.py only): measurement builds WITHOUT round 4's two fences against the co-residency bug
# (_build.BUG_LIB) — plain, with gemm_w1's MFMAs as compiler builtins (scripts/coresidency/build_bug2.sh), with parts of gemm_w1's loop removed
# (scripts/coresidency/build_bug_strips.sh)
LIB_PATH = (os.path.join(os.path.dirname(HERE), "scripts", "probes", f"libfvk_{_PROBE_SEL}.so" if _PROBE_SEL.startswith("bug") else "libfvk_probe.so")
            if PROBE else os.path.join(HERE, "libfvk_amd.so"))
ABI_VERSION = 8

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_long, C.c_float


class AttnArgs(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("vt", vp), ("o", vp), ("lse", vp),
                ("B", i32), ("H", i32), ("Sq", i32), ("Skv", i32), ("Skv_pad", i32),
                ("q_bs", i64), ("q_ss", i64), ("q_hs", i64), ("k_bs", i64), ("k_ss", i64), ("k_hs", i64),
                ("o_bs", i64), ("o_ss", i64), ("o_hs", i64), ("scale", f32), ("qk_dim", i32)]


# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "fvk_last_error": [],
    "fvk_abi_version": [],
    "fvk_is_probe_build": [],
    "fvk_device_arch": [C.c_char_p, i32],
    "fvk_set_tunable": [C.c_char_p, i32],
    "fvk_ln_modulate_bf16": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp],
    "fvk_ln_modulate_fp8_bf16": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp],
    "fvk_scale_residual_bf16": [vp, vp, vp, vp, i32, i32, i32, vp],
    "fvk_rmsnorm_rope_bf16": [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, vp, vp, i32, i32, i32, i32, i32, i64, i64, f32, vp],
    "fvk_rmsnorm_rope_scatter_bf16": [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), i32, vp, vp, i32, i32, i32, i32, i32, i64, i64, f32,
                                      C.POINTER(vp), vp],
    "fvk_qkv_norm_rope_pack_bf16": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i64, i32, i32, f32, vp],
    "fvk_qkv_norm_rope_pack2_bf16": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i64, i32, i32, f32, vp],
    "fvk_qkvg_norm_rope_pack_bf16": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i64, i32, i32, f32, vp],
    "fvk_v_transpose_bf16": [vp, vp, i32, i32, i32, i32, i64, i64, i64, i32, vp],
    "fvk_v_transpose_gather_bf16": [vp, vp, vp, i32, i32, i32, i32, i64, i64, i64, i32, vp],
    "fvk_gemm_bf16": [vp, vp, vp, vp, i32, i32, i32, i64, i64, i32, vp, vp, i32, vp],
    "fvk_gemm_bf16_batched": [vp, vp, vp, i32, i32, i32, i32, i64, i64, i64, i64, i64, i32, f32, vp],
    "fvk_gemm_vt_bf16": [vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, i32, vp],
    "fvk_mfma_sustained_probe_bf16": [vp, i32, i32, i32, vp],
    "fvk_attn_dense_bf16": [C.POINTER(AttnArgs), vp],
    "fvk_attn_dense_kernel_bf16": [C.POINTER(AttnArgs), i32, vp],
    "fvk_attn_dense_split_bf16": [C.POINTER(AttnArgs), i32, vp, vp, vp],
    "fvk_attn_block_sparse_bf16": [C.POINTER(AttnArgs), vp, vp, vp, i32, i32, vp],
    "fvk_attn_block_sparse_ws_bf16": [C.POINTER(AttnArgs), vp, vp, vp, i32, i32, vp, i64, vp],
    "fvk_attn_block_sparse_workspace_bytes": [C.POINTER(AttnArgs), i32, i32],
    "fvk_vsa_union_lists": [vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "fvk_attn_block_sparse_union_bf16": [C.POINTER(AttnArgs), vp, vp, i32, vp],
    "fvk_attn_tile_lists_bf16": [C.POINTER(AttnArgs), vp, vp, vp, i32, i32, vp, vp, vp],
    "fvk_attn_sta_bf16": [C.POINTER(AttnArgs), i32, i32, i32, i32, C.POINTER(C.c_int32), vp],
    "fvk_vsa_build_metadata_host": [i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp],
    "fvk_gather_rows_bf16": [vp, vp, vp, vp, i32, i32, i32, i64, i64, vp],
    "fvk_gather_rows_strided_bf16": [vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, i64, vp],
    "fvk_block_mean_bf16": [vp, vp, vp, i32, i32, i32, i32, i32, i64, i64, i64, vp],
    "fvk_block_mean_gather_bf16": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i64, i64, i64, vp],
    "fvk_topk_mask": [vp, i32, vp, i32, i32, i32, vp],
    "fvk_map_to_index": [vp, vp, vp, i32, i32, vp],
    "fvk_softmax_rows_bf16": [vp, vp, i32, i32, vp],
    "fvk_vsa_combine_bf16": [vp, vp, vp, vp, i32, i32, i32, i32, i32, i64, i64, i64, vp],
    "fvk_vsa_combine_scatter_bf16": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i64, i64, i64, i64, i64, i64, i64, i64, i64, vp],
    "fvk_patchify_bf16": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "fvk_unpatchify_bf16": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "fvk_timestep_embedding_bf16": [vp, vp, i32, i32, f32, vp],
    "fvk_silu_bf16": [vp, vp, i64, vp],
    "fvk_fp8_quantize_bf16": [vp, vp, vp, vp, i32, i32, i64, i32, vp],
    "fvk_gemm_fp8": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i64, i32, i32, i32, vp, vp, i32, vp],
    "fvk_cfg_unipc_step": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, C.POINTER(C.c_float), i32, i32, vp],
    "fvk_dmd_step": [vp, vp, i32, vp, vp, vp, vp, vp, i64, i64, vp],
    "fvk_vae_conv_bf16": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i64, i64, i64, i32, i32, vp],
    "fvk_vae_conv_norm_bf16": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i64, i64, i32, vp, vp, i32, i32, i32, vp],
    "fvk_vae_rmsnorm_silu_bf16": [vp, vp, vp, i64, i32, i32, i32, i32, i32, vp],
    "fvk_vae_blend_f32": [vp, vp, i64, i64, i64, i32, i32, i32, i64, i64, i64, i64, i64, i64, vp],
    "fvk_vae_postprocess_u8": [vp, vp, i32, i32, i32, i64, vp],
}
_RESTYPES = {"fvk_last_error": C.c_char_p, "fvk_attn_block_sparse_workspace_bytes": C.c_long}

_lib = None


def load() -> C.CDLL:
    """Load (once) and type the library.  Raises RuntimeError when it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950).  fastvideo_amd has no CPU / eager fallback.")
    # torch FIRST: it brings the HIP runtime the process will use for device memory and streams.  Loading this library before torch
    # (e.g. `build()` then `smoke()` in one fresh process) binds it to a second copy of the runtime, whose first launch then fails with
    # "no ROCm-capable device is detected" (seen on the GPU box; tests/test_gpu_model.py::test_build_then_smoke_in_a_fresh_process).
    import torch  # noqa: F401
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    if bool(lib.fvk_is_probe_build()) != PROBE:
        raise RuntimeError(f"{LIB_PATH}: probe-build flag {lib.fvk_is_probe_build()} does not match FVK_PROBE_LIB={int(PROBE)}; rebuild it")
    if lib.fvk_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libfvk_amd.so ABI {lib.fvk_abi_version()} != expected {ABI_VERSION}; rebuild it")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().fvk_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


_TRACE = os.environ.get("FVK_TRACE_CALLS") == "1"   # fault hunts (scripts/guard_sweep.sh): name every call, run it to completion


def call(name: str, *args) -> None:
    if _TRACE:
        import sys
        import torch
        print(f"[fvk] {name}", file=sys.stderr, flush=True)
        check(getattr(load(), name)(*args), name)
        torch.cuda.synchronize()
        return
    check(getattr(load(), name)(*args), name)
