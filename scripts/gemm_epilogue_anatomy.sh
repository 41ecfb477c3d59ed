#!/bin/bash
# GEMM epilogue anatomy: rebuild the measurement library on the box with -DW1_ABL=n and time the shipped variant (125)
cd /root/repo
for abl in 0 3; do
  FVK_EXTRA_FLAGS="-DW1_ABL=$abl" python -c "
from fastvideo_amd import _build
_build.build_probe(force=True, verbose=False)" > /dev/null 2>&1
  echo "== W1_ABL=$abl"; FVK_PROBE_LIB=1 timeout 300 python scripts/gemm_ab.py 125 2>&1 | grep -v amdgpu | cut -c1-120
done
