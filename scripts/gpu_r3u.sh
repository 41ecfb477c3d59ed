#!/bin/bash
mkdir -p gpurun_out/r3u; cd /root/repo
FVK_PROBE_LIB=1 timeout 600 python -m pytest scripts/probes/variant_tests.py -q -k "gemm" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -q 2>&1 | tail -4
KERNEL=attn_w64 PASS_TIMEOUT=100 bash scripts/pmc_traffic.sh r3u > gpurun_out/r3u/pmc_w64.log 2>&1; tail -4 gpurun_out/r3u/pmc_w64.log
cp gpurun_out/pmc/r3u/pmc_attn_w64.json profiles/r03u_pmc_attn_w64.json 2>/dev/null
PASS_TIMEOUT=100 bash scripts/pmc_traffic.sh r3u16 > gpurun_out/r3u/pmc_w16.log 2>&1; tail -4 gpurun_out/r3u/pmc_w16.log
for m in fp8 fp8_channel; do timeout 600 python bench.py --quant $m --no-cpu-baseline --no-vae 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$m', j['ms_per_step'], j['roofline'].get('kernel_choice',{}).get('kept'))"; done
timeout 600 python bench.py --no-cpu-baseline --no-vae 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('bf16', j['ms_per_step'], j['roofline'].get('kernel_choice',{}).get('kept'), j['roofline']['traffic'])"
