#!/bin/bash
mkdir -p gpurun_out/r3t; cd /root/repo
timeout 600 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_model.py -m gpu -q -k "fp8" 2>&1 | tail -6
timeout 600 python scripts/fp8_bench.py 2>&1 | tail -6 | tee gpurun_out/r3t/fp8_bench.log
