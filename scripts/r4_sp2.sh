#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r4sp2
timeout 1500 python -m pytest tests/test_gpu_sp.py tests/test_gpu_bench.py -q -k "pipelined or two_ranks" > gpurun_out/r4sp2/tests.log 2>&1; echo rc=$?; tail -5 gpurun_out/r4sp2/tests.log | cut -c1-400
