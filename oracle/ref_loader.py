"""Import harness for the *real* reference (hao-ai-lab/FastVideo) on CPU.

TEST INFRASTRUCTURE ONLY.  Nothing under ``fastvideo_amd/`` may import this
module; it is used by ``oracle/make_golden.py`` (fixture generation, run in the
build container where ``/root/reference`` is mounted), by the ``-m "not gpu"``
tests that pin ``oracle/wan_oracle.py`` against the reference itself, and by
``bench.py: cpu_baseline`` (kind "reference").  On the GPU box ``/root/reference``
does not exist; the git-ignored copy staged by ``oracle/stage_ref.py`` under
``oracle/_ref/reference`` is used when present, else :func:`available` is False.

Recipe = SURVEY.md Appendix A:
  * pre-register an empty ``fastvideo`` package so ``fastvideo/__init__.py``
    (which imports the whole product) is skipped,
  * stub the 4 missing third-party modules on the hot path,
  * give ``CpuPlatform`` an attention backend (reference ``platforms/cpu.py:11-53``
    has none; ``platforms/mps.py:49-56`` shows the intended one-liner),
  * make ``get_local_torch_device`` return cpu (``distributed/parallel_state.py:881-890``).
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys
import types

_STAGED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "reference")  # oracle/stage_ref.py's copy (GPU box)


def _default_root() -> str:
    if "FVK_REFERENCE_ROOT" in os.environ:
        return os.environ["FVK_REFERENCE_ROOT"]
    if os.path.isdir("/root/reference/fastvideo"):
        return "/root/reference"
    return _STAGED


REF_ROOT = _default_root()
_STUBS = {"imageio", "torchvision", "diffusers", "remote_pdb"}
_state = {"installed": False, "dist": False}


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "fastvideo"))


class _Stub(types.ModuleType):

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return type(n, (), {
            "__init__": lambda s, *a, **k: None,
            "__call__": lambda s, *a, **k: None,
            "__class_getitem__": classmethod(lambda c, k: c),
        })


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in _STUBS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, m):
        pass


def install() -> None:
    """Make ``import fastvideo.<hot-path module>`` work on CPU."""
    if _state["installed"]:
        return
    if not available():
        raise RuntimeError(f"reference checkout not found at {REF_ROOT}")
    missing = {m for m in _STUBS if importlib.util.find_spec(m) is None}
    _STUBS.intersection_update(missing)
    sys.meta_path.insert(0, _Finder())
    pkg = types.ModuleType("fastvideo")
    pkg.__path__ = [os.path.join(REF_ROOT, "fastvideo")]
    sys.modules["fastvideo"] = pkg
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    from fastvideo.platforms.cpu import CpuPlatform
    CpuPlatform.get_attn_backend_cls = classmethod(
        lambda c, sel, hs, dt: "fastvideo.attention.backends.sdpa.SDPABackend")
    # The harness runs the reference on the HOST cores everywhere — also on the GPU box, where the reference's platform probe
    # (fastvideo/platforms/__init__.py:117-157, amdsmi) could otherwise resolve to RocmPlatform: pin the CPU platform.
    import fastvideo.platforms as _P
    _P._current_platform = CpuPlatform()
    import torch
    import fastvideo.distributed.parallel_state as ps
    ps.get_local_torch_device = lambda: torch.device("cpu")
    _state["installed"] = True


def init_distributed(sp_size: int | None = None) -> None:
    """gloo world; SP group spans the whole world (reference parallel_state.py:893-924)."""
    install()
    if _state["dist"]:
        return
    from fastvideo.distributed import maybe_init_distributed_environment_and_model_parallel as init
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    init(1, sp_size or ws)
    _state["dist"] = True


def load_kernel_module(rel: str, name: str):
    """Load a standalone file from fastvideo-kernel by path (e.g. vsa_utils.py)."""
    path = os.path.join(REF_ROOT, "fastvideo-kernel", rel)
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build_wan(num_heads=12, head_dim=128, ffn_dim=8960, num_layers=30, text_dim=4096, seed=0,
              dtype=None, modulation_std=0.0):
    """Reference ``WanTransformer3DModel`` with the deterministic init of
    ``fastvideo/tests/distributed/test_sp_wan.py:113-126`` (Xavier for >=2-D params,
    norm weights 1, other 1-D params 0); ``modulation_std>0`` additionally draws biases /
    scale_shift tables ~N(0, std) so that the AdaLN path is exercised."""
    import torch
    init_distributed()
    from fastvideo.configs.models.dits.wanvideo import WanVideoArchConfig, WanVideoConfig
    from fastvideo.models.dits.wanvideo import WanTransformer3DModel
    cfg = WanVideoConfig(arch_config=WanVideoArchConfig(num_attention_heads=num_heads,
                                                        attention_head_dim=head_dim,
                                                        ffn_dim=ffn_dim,
                                                        num_layers=num_layers,
                                                        text_dim=text_dim))
    model = WanTransformer3DModel(config=cfg, hf_config={}).float().eval()
    init_wan_params(model, seed=seed, modulation_std=modulation_std)
    if dtype is not None:
        model = model.to(dtype)
    return model


def build_causal_wan(num_heads=12, head_dim=128, ffn_dim=8960, num_layers=30, text_dim=4096, seed=0, dtype=None, modulation_std=0.0,
                     local_attn_size=-1, sink_size=0, num_frames_per_block=3, rope_cache_policy="absolute"):
    """Reference ``CausalWanTransformer3DModel`` (fastvideo/models/dits/causal_wanvideo.py:345-417), same deterministic init as build_wan."""
    import torch  # noqa: F401
    init_distributed()
    from fastvideo.configs.models.dits.wanvideo import WanVideoArchConfig, WanVideoConfig
    from fastvideo.models.dits.causal_wanvideo import CausalWanTransformer3DModel
    cfg = WanVideoConfig(arch_config=WanVideoArchConfig(num_attention_heads=num_heads, attention_head_dim=head_dim, ffn_dim=ffn_dim,
                                                        num_layers=num_layers, text_dim=text_dim, local_attn_size=local_attn_size,
                                                        sink_size=sink_size, num_frames_per_block=num_frames_per_block,
                                                        rope_cache_policy=rope_cache_policy))
    model = CausalWanTransformer3DModel(config=cfg, hf_config={}).float().eval()
    init_wan_params(model, seed=seed, modulation_std=modulation_std)
    if dtype is not None:
        model = model.to(dtype)
    return model


def init_wan_params(model, seed=0, modulation_std=0.0) -> None:
    import torch
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() >= 2 and "scale_shift_table" not in name:
                fan_out, fan_in = p.shape[0], p[0].numel()
                bound = (6.0 / (fan_in + fan_out))**0.5
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * bound)
            elif "norm" in name and name.endswith("weight"):
                p.fill_(1.0)
                if modulation_std > 0:
                    p.add_(torch.randn(p.shape, generator=g) * modulation_std)
            else:
                p.zero_()
                if modulation_std > 0:
                    p.add_(torch.randn(p.shape, generator=g) * modulation_std)


def load_unipc_scheduler():
    """The REAL reference FlowUniPCMultistepScheduler class (fastvideo/models/schedulers/scheduling_flow_unipc_multistep.py),
    importable on CPU through oracle/_diffusers_shim.py (diffusers itself is not installed here)."""
    install()
    from oracle import _diffusers_shim
    for n in list(sys.modules):  # replace the generic attribute stubs by the functional shim
        if n == "diffusers" or n.startswith("diffusers."):
            del sys.modules[n]
    _diffusers_shim.install()
    mod = importlib.import_module("fastvideo.models.schedulers.scheduling_flow_unipc_multistep")
    return mod.FlowUniPCMultistepScheduler
