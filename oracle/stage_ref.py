"""Stage the REAL reference's hot-path Python sources into the git-ignored ``oracle/_ref/reference/`` tree.

TEST / BASELINE INFRASTRUCTURE ONLY — nothing under ``fastvideo_amd/`` imports this or anything it stages.

Why: ``/root/reference`` exists only in the build container.  ``oracle/_ref/`` is listed in ``.gitignore`` (the reference's
sources never enter this repository's history) but NOT in ``.gpurunignore``, so the staged copy travels to the GPU box with the
snapshot, like the built ``libfvk_amd.so``.  There it serves two purposes:

  * ``bench.py: cpu_baseline`` times the reference's own ``WanTransformer3DModel`` (SURVEY App. A harness) on the GPU box's host
    cores (``kind: "reference"``), the port (``oracle/wan_oracle.py``) stays as the fallback when the staged tree is absent;
  * ``tests/test_gpu_ref_triton.py`` runs the reference's own Triton VSA / STA kernels (Triton-HIP) next to the HIP
    kernels as a GPU-side checker: ``block_sparse_attn_triton.py``, ``fused_compress_topk.py``, ``index.py``,
    ``st_attn_triton.py``.

What is copied: the reference modules listed in the committed ``oracle/ref_files.json`` (paths only) — the modules the App. A
import recipe loads for a DiT forward, a VAE decode, the VSA backend and the UniPC scheduler, plus the four Triton kernel files and
``vsa_utils.py`` of the kernel wheel.  Files are copied byte for byte; ``MANIFEST.json`` records path + sha256 and a digest of the
whole tree (``tree_sha256``), which ``bench.py`` prints next to a ``kind: "reference"`` CPU baseline.

``stage()`` is a PURE FILE COPY: it never imports or executes reference code.  It is an explicit step of the test / bench harness
(``python -m oracle.stage_ref``, also called by ``scripts/gpu_round.sh`` before a gpurun) and ``__graft_entry__.build()`` calls it
best-effort AFTER the product library is built and loaded (a failure there is a warning, never a build failure).  The list itself is regenerated only on request (``python -m oracle.stage_ref --discover``: runs the import recipe in a
subprocess and lists ``sys.modules`` — the one place reference code executes, never part of a build).
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = "/root/reference"
DST = os.path.join(HERE, "_ref", "reference")
MANIFEST = os.path.join(HERE, "_ref", "MANIFEST.json")

KERNEL_FILES = [
    "fastvideo-kernel/python/fastvideo_kernel/triton_kernels/block_sparse_attn_triton.py",
    "fastvideo-kernel/python/fastvideo_kernel/triton_kernels/fused_compress_topk.py",
    "fastvideo-kernel/python/fastvideo_kernel/triton_kernels/index.py",
    "fastvideo-kernel/python/fastvideo_kernel/triton_kernels/st_attn_triton.py",
    "fastvideo-kernel/python/fastvideo_kernel/vsa_utils.py",
]

_PROBE = r"""
import json, os, sys
sys.path.insert(0, {root!r})
os.environ["FVK_REFERENCE_ROOT"] = {src!r}
import torch
from oracle import ref_loader as R
R.install()
m = R.build_wan(num_heads=2, head_dim=128, ffn_dim=256, num_layers=1, text_dim=64)
from fastvideo.forward_context import set_forward_context
with torch.no_grad(), set_forward_context(current_timestep=0, attn_metadata=None):
    m(hidden_states=torch.randn(1, 16, 2, 4, 4), encoder_hidden_states=torch.randn(1, 8, 64), timestep=torch.tensor([500]))
R.build_causal_wan(num_heads=2, head_dim=128, ffn_dim=256, num_layers=1, text_dim=64)
import fastvideo.models.vaes.wanvae, fastvideo.configs.models.vaes.wanvae, fastvideo.models.vaes.common
import fastvideo.attention.backends.video_sparse_attn, fastvideo.attention.backends.sdpa, fastvideo.attention.selector
import fastvideo.layers.quantization.fp8_config, fastvideo.layers.custom_op, fastvideo.layers.linear, fastvideo.layers.layernorm
import fastvideo.layers.rotary_embedding, fastvideo.models.utils, fastvideo.platforms.rocm, fastvideo.platforms.interface
try:
    import fastvideo.layers.quantization
except Exception:
    pass
try:
    R.load_unipc_scheduler()
    import fastvideo.models.schedulers.scheduling_flow_match_euler_discrete
except Exception:
    pass
files = sorted({{getattr(mod, "__file__", None) for mod in list(sys.modules.values())
                if getattr(mod, "__file__", None) and str(mod.__file__).startswith({src!r} + "/")}})
print("FILES=" + json.dumps(files))
"""


def _sha(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def available() -> bool:
    return os.path.isdir(os.path.join(SRC, "fastvideo"))


def staged_root() -> str | None:
    """Path of the staged tree if it exists (what ``oracle/ref_loader.py`` falls back to on the GPU box)."""
    return DST if os.path.isdir(os.path.join(DST, "fastvideo")) else None


FILE_LIST = os.path.join(HERE, "ref_files.json")


def discover() -> list[str]:
    """Explicit opt-in: run the App. A import recipe in a subprocess (this EXECUTES reference code) and rewrite ref_files.json."""
    env = dict(os.environ, MASTER_PORT=os.environ.get("FVK_STAGE_PORT", "29533"), WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, "-c", _PROBE.format(root=ROOT, src=SRC)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, env=env, cwd=ROOT)
    line = [l for l in r.stdout.splitlines() if l.startswith("FILES=")]
    if r.returncode != 0 or not line:
        raise RuntimeError(f"stage_ref: the reference import probe failed:\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}")
    files = sorted(set([os.path.relpath(f, SRC) for f in json.loads(line[0][6:])] + KERNEL_FILES))
    old = json.load(open(FILE_LIST)) if os.path.exists(FILE_LIST) else {}
    json.dump({"comment": old.get("comment", "paths copied by oracle/stage_ref.py"), "files": files}, open(FILE_LIST, "w"), indent=1)
    return files


def manifest() -> dict | None:
    """The manifest of the staged tree ({"files": {path: sha256}, "tree_sha256": ...}) or None."""
    try:
        return json.load(open(MANIFEST)) if staged_root() else None
    except Exception:  # noqa: BLE001
        return None


def stage(force: bool = False, verbose: bool = False) -> bool:
    if not available():
        return False
    if not force and os.path.exists(MANIFEST):
        try:
            man = json.load(open(MANIFEST))
            if all(os.path.exists(os.path.join(SRC, f)) and _sha(os.path.join(SRC, f)) == h and os.path.exists(os.path.join(DST, f))
                   for f, h in man["files"].items()):
                return True
        except Exception:  # noqa: BLE001 - a damaged manifest just means: stage again
            pass
    files = json.load(open(FILE_LIST))["files"]
    if os.path.isdir(DST):
        shutil.rmtree(DST)
    man = {}
    for rel in sorted(set(files)):
        src, dst = os.path.join(SRC, rel), os.path.join(DST, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(src, dst)
        man[rel] = _sha(src)
    os.makedirs(os.path.dirname(MANIFEST), exist_ok=True)
    tree = hashlib.sha256("".join(f"{k}:{v}\n" for k, v in sorted(man.items())).encode()).hexdigest()
    json.dump({"source": SRC, "files": man, "tree_sha256": tree}, open(MANIFEST, "w"), indent=0)
    if verbose:
        print(f"stage_ref: {len(man)} reference files -> {DST}")
    return True


if __name__ == "__main__":
    if "--discover" in sys.argv:
        print(f"discovered {len(discover())} reference files -> {FILE_LIST}")
    ok = stage(force="--force" in sys.argv, verbose=True)
    print("staged" if ok else "reference checkout not present: nothing staged")
