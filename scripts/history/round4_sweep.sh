#!/bin/bash
# round 4, last visit: the bench lines of the configurations beside the contract one, final code, ONE box (so that they can be compared with each other)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4sweep
mkdir -p "$OUT"
run() { name=$1; shift; timeout 900 python bench.py "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; echo "$name rc=$? $(python -c "
import json,sys
try:
    j=json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1]); r=j['roofline']
    print('ms/step', j['ms_per_step'], 'step TF', j.get('step_tflops'), '| roofline', r['achieved'], r['frac'], '| vae', (j.get('vae') or {}).get('ms_per_step'), '| sclk', ((j.get('power') or {}).get('sclk_mhz') or {}).get('p50'))
except Exception as e: print('parse error', e)
")"; }
run contract --no-cpu-baseline --no-cfg-step
run sta --attention sta --no-cpu-baseline --no-vae --no-cfg-step
run vsa --attention vsa --no-cpu-baseline --no-vae --no-cfg-step
run fp8 --quant fp8 --no-cpu-baseline --no-vae --no-cfg-step
run fp8c --quant fp8_channel --no-cpu-baseline --no-vae --no-cfg-step
run cfg1 --config cfg1 --no-cpu-baseline --no-cfg-step
run cfg5 --config cfg5 --no-cpu-baseline --no-cfg-step --steps 3 --warmup 1
run cfg5_vsa_fp8 --config cfg5 --attention vsa --quant fp8 --no-cpu-baseline --no-vae --no-cfg-step --steps 3 --warmup 1
run cfg4 --config cfg4 --no-cpu-baseline --no-vae --no-cfg-step --steps 2 --warmup 1
