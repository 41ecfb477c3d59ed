"""Minimal stand-ins for the few `diffusers` names the reference's FlowUniPCMultistepScheduler imports
(fastvideo/models/schedulers/scheduling_flow_unipc_multistep.py:11-13), so that the REAL reference scheduler can run on CPU in
this container (diffusers is not installed).  TEST INFRASTRUCTURE ONLY: used by oracle/ref_loader.load_unipc_scheduler() to
generate / pin the scheduler golden vectors; nothing here restates scheduler arithmetic."""
from __future__ import annotations

import enum
import functools
import inspect
import sys
import types
from dataclasses import dataclass


class _Config(dict):
    __getattr__ = dict.__getitem__


class ConfigMixin:
    config_name = "scheduler_config.json"

    def register_to_config(self, **kwargs):
        if not hasattr(self, "_internal_dict"):
            self._internal_dict = _Config()
        self._internal_dict.update(kwargs)

    @property
    def config(self):
        return self._internal_dict


def register_to_config(init):
    """diffusers.configuration_utils.register_to_config: record the (defaulted) __init__ arguments in self.config."""
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        params = [p for n, p in sig.parameters.items() if n != "self" and p.kind not in (p.VAR_KEYWORD, p.VAR_POSITIONAL)]
        cfg = {p.name: p.default for p in params}
        cfg.update({p.name: a for p, a in zip(params, args)})
        cfg.update({k: v for k, v in kwargs.items() if k in cfg})
        self.register_to_config(**cfg)
        init(self, *args, **kwargs)
    return inner


class SchedulerMixin:
    pass


class KarrasDiffusionSchedulers(enum.Enum):
    UniPCMultistepScheduler = 1


@dataclass
class SchedulerOutput:
    prev_sample: object


def deprecate(*args, **kwargs):
    return None


class BaseOutput(dict):
    """diffusers.utils.BaseOutput stand-in (scheduling_flow_match_euler_discrete.py:28 subclasses it for its step output)."""


def install() -> None:
    pkg = types.ModuleType("diffusers"); pkg.__path__ = []
    cu = types.ModuleType("diffusers.configuration_utils"); cu.ConfigMixin, cu.register_to_config = ConfigMixin, register_to_config
    sch = types.ModuleType("diffusers.schedulers"); sch.__path__ = []
    su = types.ModuleType("diffusers.schedulers.scheduling_utils")
    su.KarrasDiffusionSchedulers, su.SchedulerMixin, su.SchedulerOutput = KarrasDiffusionSchedulers, SchedulerMixin, SchedulerOutput
    ut = types.ModuleType("diffusers.utils"); ut.deprecate = deprecate; ut.BaseOutput = BaseOutput; ut.__path__ = []  # package: other submodules fall through to the generic stubs
    for n, m in (("diffusers", pkg), ("diffusers.configuration_utils", cu), ("diffusers.schedulers", sch),
                 ("diffusers.schedulers.scheduling_utils", su), ("diffusers.utils", ut)):
        sys.modules.setdefault(n, m)
