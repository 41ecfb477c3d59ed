"""Build libfvk_amd.so (hipcc, gfx950) in-tree.  Called by ``__graft_entry__.build()`` (and by ``bench.py`` through it).
``fastvideo_amd._lib.load()`` never builds: a missing library is an error there (no silent slow path).
The built .so lives next to the sources so that it travels with the repo snapshot to the GPU box."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libfvk_amd.so")
SOURCES = ["capi.hip", "norm_mod.hip", "gemm_bf16.hip", "gemm_pp.hip", "gemm_ph.hip", "gemm_w1.hip", "gemm_w1n.hip", "fp8.hip", "attn_fwd.hip", "attn_bs16.hip", "attn_pp2.hip", "attn_w16.hip", "attn_w64.hip", "vsa_misc.hip", "vae_conv.hip", "vae_conv3.hip", "vae_conv3w.hip", "vae_convout.hip", "vae_post.hip", "sched_step.hip", "mfma_probe.hip"]
# measurement build (scripts/probes/libfvk_probe.so): the same sources with -DFVK_PROBE_BUILD (variant dispatch + fvk_set_tunable knobs
# compiled in) plus the experiment kernels that never shipped.  Nothing in the product path loads it (fastvideo_amd/_lib.py: FVK_PROBE_LIB=1).
PROBE_DIR = os.path.join(HERE, "..", "scripts", "probes")
PROBE_SOURCES = ["attn_pp.hip", "attn_vsa.hip"]
PROBE_LIB = os.path.join(PROBE_DIR, "libfvk_probe.so")
# round 5, scripts/coresidency/coresidency_matrix.py only: the measurement build WITHOUT the two fences of round 4's co-residency bug (the register-file
# claim of the one-wave-per-SIMD kernels, the packed-fp32 ban in the small kernels' files) — the pre-fix state, rebuilt on purpose to study it
BUG_LIB = os.path.join(PROBE_DIR, "libfvk_bug.so")
# The files of the SMALL kernels (norm / RoPE / pack, gathers, quantisers, scheduler step, post-processing: few registers, so their waves can
# share a SIMD with another kernel's) are compiled WITHOUT packed-fp32 VALU instructions: round 4 found v_pk_mul_f32 / v_pk_add_f32 results of
# such a wave wrong while a foreign MFMA stream ran on its SIMD (profiles/r04z_pk_f32_beside_mfma.log).  The known aggressors now
# claim the whole register file; this is the second fence, for neighbours we do not control.  Same arithmetic (two scalar operations instead of
# one packed one), and these kernels are HBM-bound.  (The host pass prints "not a recognized feature" for the flag: filtered below.)
NO_PACKED_FP32 = ("norm_mod.hip", "vsa_misc.hip", "fp8.hip", "sched_step.hip", "vae_post.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]
FLAGS += os.environ.get("FVK_EXTRA_FLAGS", "").split()  # measurement builds only (e.g. -DFVK_ST_ABL=1 timing ablations)


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libfvk_amd.so")
    return exe


def _flag_stamp(probe: bool) -> str:
    """What the library was built WITH: the flag list (FVK_EXTRA_FLAGS included) and this file (the per-file extras — unroll thresholds, the
    no-packed-fp32 fence — live in its code).  Stored beside the library; a library whose stamp differs is stale even if it is newer than every
    source (a flag-only change such as the packed-fp32 fence must not be skipped silently)."""
    import hashlib
    h = hashlib.sha256()
    h.update(repr((FLAGS, NO_PACKED_FP32, SOURCES, PROBE_SOURCES if probe else None)).encode())
    with open(os.path.abspath(__file__).replace(".pyc", ".py"), "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def needs_build(probe: bool = False) -> bool:
    lib = PROBE_LIB if probe else LIB
    if not os.path.exists(lib):
        return True
    try:
        with open(lib + ".flags") as f:
            if f.read().strip() != _flag_stamp(probe):
                return True
    except OSError:
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f != "build"] + [os.path.join(HERE, "..", "include", "fvk_amd.h")]
    if probe:
        deps += [os.path.join(PROBE_DIR, f) for f in PROBE_SOURCES]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, probe: bool = False) -> str:
    lib = PROBE_LIB if probe else LIB
    if not force and not needs_build(probe):
        return lib
    # one builder at a time: under torch.distributed.run every rank calls build(), and a snapshot copy may have refreshed mtimes
    import fcntl
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build(probe):  # another rank finished the build while this one waited
                return lib
            return _build_locked(verbose, probe)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def build_probe(force: bool = False, verbose: bool = True) -> str:
    """The measurement build (see PROBE_SOURCES)."""
    return build(force, verbose, probe=True)


GUARD_LIB = os.path.join(PROBE_DIR, "libguard_alloc.so")


def build_guard(force: bool = False) -> str:
    """scripts/probes/libguard_alloc.so: the guard-page device allocator of the out-of-bounds hunts (tests/_guard.py, tests/test_gpu_guard.py).
    Host code only; test infrastructure, never loaded by the product."""
    src = os.path.join(PROBE_DIR, "guard_alloc.cpp")
    if not force and os.path.exists(GUARD_LIB) and os.path.getmtime(GUARD_LIB) >= os.path.getmtime(src):
        return GUARD_LIB
    r = subprocess.run([hipcc(), "-O2", "-shared", "-fPIC", src, "-o", GUARD_LIB], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on guard_alloc.cpp:\n{r.stdout}")
    return GUARD_LIB


def build_bug(verbose: bool = True) -> str:
    """scripts/probes/libfvk_bug.so (see BUG_LIB); always rebuilt, never built by __graft_entry__.build()."""
    return _build_locked(verbose, probe=True, bug=True)


def _build_locked(verbose: bool, probe: bool = False, bug: bool = False) -> str:
    objdir = os.path.join(PROBE_DIR, "build_bug" if bug else "build") if probe else os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    objs = []
    procs = []
    lib = BUG_LIB if bug else PROBE_LIB if probe else LIB
    srcs = [(src, os.path.join(CSRC, src)) for src in SOURCES] + ([(src, os.path.join(PROBE_DIR, src)) for src in PROBE_SOURCES] if probe else [])
    for src, path in srcs:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        extra = ["-ffp-contract=off"] if src in ("sched_step.hip", "vae_post.hip") else []  # bit-exact fp32 arithmetic (no fused multiply-add)
        if src in ("attn_w16.hip", "attn_w64.hip", "attn_bs16.hip", "gemm_w1.hip", "gemm_w1n.hip", "vae_conv3w.hip"):
            # its 64-chunk iteration must be FULLY unrolled (every register-array index a constant): above clang's default budget for
            # `#pragma unroll`, silently left as a loop otherwise — with the wave's whole register struct in scratch
            extra += ["-mllvm", "-pragma-unroll-threshold=100000"]
        if src in NO_PACKED_FP32 and not bug:
            extra += ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
        if probe:
            extra += ["-DFVK_PROBE_BUILD=1", "-I", CSRC]
        if bug:
            extra += ["-DFVK_NO_REGISTER_CLAIM=1"]
        cmd = [cc, *FLAGS, *extra, "-c", path, "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        out = "\n".join(ln for ln in out.splitlines() if "'-packed-fp32-ops' is not a recognized feature" not in ln)
        if verbose and out.strip():
            print(out)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    import ctypes
    import torch  # noqa: F401  (first: the process must end up with ONE HIP runtime — torch's; see fastvideo_amd/_lib.py load())
    try:  # catches e.g. a kernel whose host stub was not emitted (undefined symbol at dlopen time)
        ctypes.CDLL(lib)
    except OSError as e:
        os.remove(lib)
        raise RuntimeError(f"built library does not load: {e}") from e
    if not bug:
        with open(lib + ".flags", "w") as f:
            f.write(_flag_stamp(probe) + "\n")
    return lib


if __name__ == "__main__":
    print(build(force=True))
    print(build_probe(force=True))
