#!/bin/bash
# round 3: gemm_w1 (one wave per SIMD) correctness + A/B against gemm_ph and the vendor library
mkdir -p gpurun_out/r3i; cd /root/repo
export FVK_PROBE_LIB=1
timeout 600 python -m pytest scripts/probes/variant_tests.py -x -q -k "gemm" 2>&1 | tail -5 > gpurun_out/r3i/tests.log
timeout 600 python scripts/gemm_ab.py 125 253 > gpurun_out/r3i/gemm_ab.log 2>&1
cat gpurun_out/r3i/tests.log gpurun_out/r3i/gemm_ab.log
