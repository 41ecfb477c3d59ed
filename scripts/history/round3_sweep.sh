#!/bin/bash
# round 3, final visit S: FULL gpu suite, then the round's sweep of bench lines (contract, sta, vsa, fp8, cfg1, cfg5, vae)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3sweep
mkdir -p "$OUT"
echo "== full gpu suite"
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > "$OUT/pytest_full.log" 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error" "$OUT/pytest_full.log" | tail -3; grep -E "^(FAILED|ERROR)" "$OUT/pytest_full.log" | head -20
echo "== bench lines"
run() { name=$1; shift; timeout 900 python bench.py "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; echo "$name rc=$? $(python -c "
import json,sys
try:
    j=json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1]); r=j['roofline']
    print('ms/step', j['ms_per_step'], 'step TF', j.get('step_tflops'), '| roofline', r['achieved'], r['frac'], '| vae', (j.get('vae') or {}).get('ms_per_step'))
except Exception as e: print('parse error', e)
")"; }
run contract
run sta --attention sta --no-cpu-baseline --no-vae
run vsa --attention vsa --no-cpu-baseline --no-vae
run fp8 --quant fp8 --no-cpu-baseline --no-vae
run fp8c --quant fp8_channel --no-cpu-baseline --no-vae
run cfg1 --config cfg1 --no-cpu-baseline
run cfg5 --config cfg5 --no-cpu-baseline --steps 3 --warmup 1
run cfg5_vsa_fp8 --config cfg5 --attention vsa --quant fp8 --no-cpu-baseline --no-vae --steps 3 --warmup 1
run cfg4 --config cfg4 --no-cpu-baseline --steps 2 --warmup 1
