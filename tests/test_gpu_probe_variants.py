"""The non-shipping kernel variants (ping-pong schedules of attn_pp2, attn_pp.hip, attn_vsa.hip, gemm_ph variants, the lockstep VAE conv
schedule, the block-per-row top-k kernel) and the cross-kernel agreement checks live in the MEASUREMENT build of the library
(scripts/probes/libfvk_probe.so).  One process binds ONE of the two libraries (fastvideo_amd/_lib.py: FVK_PROBE_LIB), so their tests
(scripts/probes/variant_tests.py) run here in a subprocess; this process — like every other test — runs on the product library."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_measurement_build_variants():
    env = dict(os.environ, FVK_PROBE_LIB="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "scripts", "probes", "variant_tests.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = r.stdout[-3000:]
    print(tail)
    assert r.returncode == 0, tail
    assert " passed" in tail and "skipped" not in tail.splitlines()[-1], tail


@pytest.mark.gpu
def test_product_library_refuses_measurement_switches():
    """libfvk_amd.so holds the shipped configuration only: a non-zero knob is an error that names the measurement build."""
    import torch  # noqa: F401
    from fastvideo_amd import _lib, ops
    assert _lib.load().fvk_is_probe_build() == 0
    ops.set_tunable("attn_impl", 0)
    with pytest.raises(RuntimeError, match="libfvk_probe"):
        ops.set_tunable("attn_impl", 103)
    with pytest.raises(RuntimeError, match="unknown tunable"):
        ops.set_tunable("no_such_knob", 0)
