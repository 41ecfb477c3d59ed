#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4pmc
python -c "
import ctypes, os
for p in ('fastvideo_amd/libfvk_amd.so', 'scripts/probes/libfvk_probe.so'):
    ctypes.CDLL(os.path.abspath(p)); print('loads', p)
" || exit 1
timeout 600 python -m pytest tests/test_gpu_vae.py -x -q > gpurun_out/r4pmc/tests_vae.log 2>&1; echo rc=$?; tail -2 gpurun_out/r4pmc/tests_vae.log | cut -c1-300
FVK_PROBE_LIB=1 timeout 600 python -m pytest scripts/probes/variant_tests.py -x -q -k "conv3w" > gpurun_out/r4pmc/tests_variant.log 2>&1; echo rc=$?; tail -2 gpurun_out/r4pmc/tests_variant.log | cut -c1-300
bash scripts/conv_pmc_traffic.sh r4b 2>&1 | grep -i "rc=\|traffic_over\|FETCH_SIZE_KB\|WRITE_SIZE_KB"
for i in 1 2; do FVK_PROBE_LIB=1 timeout 300 python scripts/vae_conv_breakdown.py > gpurun_out/r4pmc/breakdown_$i.log 2>&1; grep -v amdgpu.ids gpurun_out/r4pmc/breakdown_$i.log | head -8 | cut -c1-200; done
FVK_PROBE_LIB=1 timeout 300 python scripts/conv_power_ab.py 2>&1 | grep "impl=0\|impl=3" | cut -c1-220
