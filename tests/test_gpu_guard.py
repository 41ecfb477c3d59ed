"""Out-of-bounds regression tests under the guard-page allocator (round 6; scripts/probes/guard_alloc.cpp, tests/_guard.py): every device tensor
ends where its mapping ends with an unmapped page behind it, so an overrun of ANY kernel is a GPU page fault that kills the process — each case
therefore runs in its own subprocess.

  * the detector detects: a deliberate 16-byte overrun must die with "Memory access fault", a correct program must stay correct;
  * the bug the first guard run found (profiles/r06d_guard_page_runs.md): the gated-residual GEMM epilogue computed its gate row from a wave
    tile's first row even when that row lay past M — for the LAST layer's FFN-out that is one row past the end of the [layers, batch, d] gate
    tensor: harmless beside a mapped neighbour (the values are never stored), a GPU fault when the tensor happens to end a mapped segment (the
    one unexplained abort of round 5's full-suite runs).  The forwards that reached it — the tiny sliding-tile model, the pipelined
    SP exchange — run here under the guard allocator;
  * the sparse paths (block lists, row maps, pad rows) of the kernel-level tests."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "scripts", "probes", "libguard_alloc.so")


def _run(args, timeout=600):
    env = dict(os.environ, FVK_GUARD_ALLOC="1")
    return subprocess.run([sys.executable, *args], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(SO):
        pytest.fail(f"{SO} not built (python -c 'import __graft_entry__ as g; g.build()')")


def test_the_detector_detects():
    ok = _run(["scripts/guard_selftest.py"])
    assert ok.returncode == 0 and "mismatches 0" in ok.stdout, ok.stdout[-2000:]
    oob = _run(["scripts/guard_selftest.py", "oob"])
    assert oob.returncode != 0 and "Memory access fault" in oob.stdout and "NO FAULT" not in oob.stdout, oob.stdout[-2000:]


@pytest.mark.parametrize("target", ["tests/test_gpu_model.py::test_wan_tiny_sta_matches_oracle",
                                    "tests/test_gpu_model.py::test_wan_tiny_forward_matches_reference",
                                    "tests/test_gpu_kernels.py"])
def test_no_out_of_bounds_access(target):
    r = _run(["-m", "pytest", target, "-x", "-q"], timeout=900)
    assert "Memory access fault" not in r.stdout, r.stdout[-3000:]
    assert r.returncode == 0, r.stdout[-3000:]
