"""Host logic of the in-place attention kernel choice (fastvideo_amd/wan_dit.py: attn_autotune) without a GPU: the decision from timed launches
(median per kernel, the first launch of each left out, ties to the default), the too-few-launches guard, and one decision for all ranks under
sequence parallelism (sum of the ranks' medians)."""
import types

import pytest
import torch

from fastvideo_amd import ops
from fastvideo_amd.wan_dit import WanTransformer3DModelHip


class _Ev:
    """stand-in for a HIP event pair: e0.elapsed_time(e1) returns the launch's duration"""

    def __init__(self, ms=None):
        self.ms = ms

    def elapsed_time(self, other):
        return other.ms


def _model(P=1, rank_sums=None):
    m = WanTransformer3DModelHip.__new__(WanTransformer3DModelHip)  # no weights, no device: only the state the decision touches
    m.attn_kernel, m.attn_autotune, m.attn_tune_report, m._tune, m.device = ops.ATTN_KERNEL_DEFAULT, True, None, [], torch.device("cpu")
    m.sp = types.SimpleNamespace(lay=types.SimpleNamespace(P=P), sum_over_ranks=lambda vals, device=None: rank_sums(vals))
    return m


def _launches(w16_ms, w64_ms):
    ev = []
    for i in range(max(len(w16_ms), len(w64_ms))):
        if i < len(w16_ms):
            ev.append((ops.ATTN_KERNEL_W16, _Ev(), _Ev(w16_ms[i])))
        if i < len(w64_ms):
            ev.append((ops.ATTN_KERNEL_W64, _Ev(), _Ev(w64_ms[i])))
    return ev


@pytest.fixture(autouse=True)
def _no_device_sync(monkeypatch):
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)


def test_the_faster_median_wins_and_the_first_launches_do_not_count():
    m = _model()
    # the first launch of each kernel is 10x slower (module load): left out; attn_w64's median is lower although one launch is an outlier
    m._tune = _launches([50.0, 5.0, 5.1, 5.0, 5.2], [60.0, 4.8, 4.9, 9.0, 4.8])
    m._finish_attn_tune()
    assert m.attn_kernel == ops.ATTN_KERNEL_W64 and m.attn_tune_report["kept"] == "attn_w64" and m.attn_tune_report["launches_timed"] == 8
    assert m._tune is None and m.attn_autotune is False
    assert m.attn_tune_report["attn_w16_ms"] == 5.1 and m.attn_tune_report["attn_w64_ms"] == 4.9
    m = _model()
    m._tune = _launches([9.0, 4.6, 4.7, 4.6], [9.0, 4.8, 4.8, 4.9])
    m._finish_attn_tune()
    assert m.attn_kernel == ops.ATTN_KERNEL_W16 and m.attn_tune_report["kept"] == "attn_w16"


def test_too_few_long_key_launches_keep_the_default():
    m = _model()
    m._tune = _launches([5.0, 5.0], [4.0, 4.0])   # 4 launches: short sequences take the 8-wave kernel, nothing to compare
    m._finish_attn_tune()
    assert m.attn_kernel == ops.ATTN_KERNEL_DEFAULT and m.attn_tune_report is None and m.attn_autotune is False


def test_sequence_parallel_ranks_take_one_decision():
    # this rank measured attn_w16 faster, the group as a whole attn_w64: every rank keeps attn_w64
    seen = {}

    def sums(vals):
        seen["vals"] = list(vals)
        return [vals[0] + 3 * 5.3, vals[1] + 3 * 4.7]   # three other ranks

    m = _model(P=4, rank_sums=sums)
    m._tune = _launches([9.0, 4.6, 4.6, 4.6], [9.0, 4.9, 4.9, 4.9])
    m._finish_attn_tune()
    assert seen["vals"] == [4.6, 4.9]
    assert m.attn_kernel == ops.ATTN_KERNEL_W64
    assert m.attn_tune_report["attn_w16_ms"] == round((4.6 + 3 * 5.3) / 4, 4) and m.attn_tune_report["attn_w64_ms"] == round((4.9 + 3 * 4.7) / 4, 4)
