#!/bin/bash
mkdir -p gpurun_out/r3w; cd /root/repo
FVK_PROBE_LIB=1 timeout 600 python -m pytest scripts/probes/variant_tests.py -q -k "gemm" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_boundary.py -m gpu -q 2>&1 | tail -3
timeout 600 python scripts/step_ab.py 2>&1 | tail -8 | cut -c1-250 | tee gpurun_out/r3w/step_ab.log
