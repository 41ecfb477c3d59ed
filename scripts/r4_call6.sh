#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4f; mkdir -p $OUT
python -c "import ctypes,torch; ctypes.CDLL('fastvideo_amd/libfvk_amd.so'); ctypes.CDLL('scripts/probes/libfvk_probe.so'); print('both libraries load')" || exit 1
echo "== vae tests"; timeout 400 python -m pytest tests/test_gpu_vae.py -q -x > $OUT/vae_tests.log 2>&1; echo rc=$?; tail -8 $OUT/vae_tests.log | cut -c1-400
echo "== conv3w vs 8-wave"; FVK_PROBE_LIB=1 timeout 300 python -m pytest scripts/probes/variant_tests.py -q -k "conv3w" > $OUT/variant.log 2>&1; echo rc=$?; tail -6 $OUT/variant.log | cut -c1-400
for impl in 0 5; do echo "== breakdown impl $impl"; FVK_PROBE_LIB=1 timeout 200 python scripts/vae_conv_breakdown.py --impl $impl > $OUT/vae_breakdown_$impl.log 2>&1; head -8 $OUT/vae_breakdown_$impl.log | cut -c1-200; done
echo "== probe impl 0"; VAE_CONV_IMPL=0 FVK_PROBE_LIB=1 timeout 300 python scripts/conv3w_probe.py > $OUT/conv3w_probe_0.log 2>&1; cat $OUT/conv3w_probe_0.log | cut -c1-420
echo "== conv power A/B"; timeout 400 python scripts/conv_power_ab.py > $OUT/conv_power_ab.log 2>&1; cat $OUT/conv_power_ab.log | cut -c1-300
