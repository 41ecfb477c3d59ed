#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4e; mkdir -p $OUT
python -c "import ctypes,torch; ctypes.CDLL('fastvideo_amd/libfvk_amd.so'); ctypes.CDLL('scripts/probes/libfvk_probe.so'); print('both libraries load')" || exit 1
echo "== sp pipe debug"; timeout 200 python scripts/sp_pipe_debug.py > $OUT/sp_pipe_debug.log 2>&1; cat $OUT/sp_pipe_debug.log | cut -c1-300
echo "== conv power A/B"; timeout 400 python scripts/conv_power_ab.py > $OUT/conv_power_ab.log 2>&1; cat $OUT/conv_power_ab.log | cut -c1-300
echo "== breakdown impl 5"; FVK_PROBE_LIB=1 timeout 200 python scripts/vae_conv_breakdown.py --impl 5 > $OUT/vae_breakdown_5.log 2>&1; head -8 $OUT/vae_breakdown_5.log | cut -c1-200
echo "== masks"; timeout 300 python -m pytest tests/test_gpu_boundary.py -q -k "key_padding" > $OUT/masks.log 2>&1; echo rc=$?; tail -8 $OUT/masks.log | cut -c1-300
