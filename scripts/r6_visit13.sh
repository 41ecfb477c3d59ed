#!/bin/bash
# visit 13: conv_out rolling three-frame kernel — tests, decode breakdown old / new, full-length parity
export TMPDIR=/tmp
O=gpurun_out/r6v13; mkdir -p $O
bash scripts/box_info.sh > $O/box_info.log 2>&1
timeout 600 python -m pytest tests/test_gpu_vae.py -m gpu -x -q > $O/vae_tests.log 2>&1; echo "vae tests rc=$?"; tail -3 $O/vae_tests.log
FVK_PROBE_LIB=1 timeout 600 python scripts/vae_conv_breakdown.py --impl 4 > $O/vae_breakdown_old_convout.log 2>&1; echo "breakdown old rc=$?"; grep -v amdgpu $O/vae_breakdown_old_convout.log | head -16
FVK_PROBE_LIB=1 timeout 600 python scripts/vae_conv_breakdown.py > $O/vae_breakdown_new_convout.log 2>&1; echo "breakdown new rc=$?"; grep -v amdgpu $O/vae_breakdown_new_convout.log | head -16
timeout 1200 python -m pytest tests/test_gpu_vae_real.py tests/test_gpu_vae_tiled.py -m gpu -x -q > $O/vae_real_tests.log 2>&1; echo "vae real tests rc=$?"; tail -3 $O/vae_real_tests.log
echo "visit 13 done"
