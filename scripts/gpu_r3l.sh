#!/bin/bash
# round 3: full GPU suite + variant tests after a kernel change
mkdir -p gpurun_out/r3l; cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_probe_variants.py 2>&1 | tail -15 > gpurun_out/r3l/tests.log
FVK_PROBE_LIB=1 timeout 900 python -m pytest scripts/probes/variant_tests.py -q 2>&1 | tail -8 > gpurun_out/r3l/variants.log
cat gpurun_out/r3l/tests.log gpurun_out/r3l/variants.log

timeout 600 python bench.py > gpurun_out/r3l/bench.json 2> gpurun_out/r3l/bench.err; tail -1 gpurun_out/r3l/bench.json | cut -c1-700
