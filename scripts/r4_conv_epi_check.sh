#!/bin/bash
# box visit: the conv3w epilogue rework (batched residual loads, counted wait past the epilogue's stores) — parity, then decode wall A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4v; mkdir -p $OUT
python -c "
import ctypes, os
for p in ('fastvideo_amd/libfvk_amd.so', 'scripts/probes/libfvk_probe.so'):
    ctypes.CDLL(os.path.abspath(p)); print('loads', p)
" || exit 1
timeout 600 python -m pytest tests/test_gpu_vae.py -x -q > $OUT/tests_vae.log 2>&1; echo rc=$?; tail -3 $OUT/tests_vae.log | cut -c1-300
FVK_PROBE_LIB=1 timeout 600 python -m pytest scripts/probes/variant_tests.py -x -q -k "conv3w" > $OUT/tests_variant.log 2>&1; echo rc=$?; tail -3 $OUT/tests_variant.log | cut -c1-300
for impl in 0 0; do
  FVK_PROBE_LIB=1 timeout 300 python scripts/vae_conv_breakdown.py --impl $impl > $OUT/breakdown_impl${impl}.log 2>&1; grep -v amdgpu.ids $OUT/breakdown_impl${impl}.log | head -8 | cut -c1-200
done
