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
SOURCES = ["capi.hip", "norm_mod.hip", "gemm_bf16.hip", "gemm_pp.hip", "gemm_ph.hip", "fp8.hip", "attn_fwd.hip", "attn_pp.hip", "attn_pp2.hip", "attn_vsa.hip", "vsa_misc.hip", "vae_conv.hip", "vae_conv3.hip", "vae_post.hip", "sched_step.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]
FLAGS += os.environ.get("FVK_EXTRA_FLAGS", "").split()  # measurement builds only (e.g. -DFVK_ST_ABL=1 timing ablations)


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libfvk_amd.so")
    return exe


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "fvk_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    # one builder at a time: under torch.distributed.run every rank calls build(), and a snapshot copy may have refreshed mtimes
    import fcntl
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():  # another rank finished the build while this one waited
                return LIB
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose: bool) -> str:
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        extra = ["-ffp-contract=off"] if src in ("sched_step.hip", "vae_post.hip") else []  # bit-exact fp32 arithmetic (no fused multiply-add)
        cmd = [cc, *FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    import ctypes
    import torch  # noqa: F401  (first: the process must end up with ONE HIP runtime — torch's; see fastvideo_amd/_lib.py load())
    try:  # catches e.g. a kernel whose host stub was not emitted (undefined symbol at dlopen time)
        ctypes.CDLL(LIB)
    except OSError as e:
        os.remove(LIB)
        raise RuntimeError(f"built library does not load: {e}") from e
    return LIB


if __name__ == "__main__":
    print(build(force=True))
