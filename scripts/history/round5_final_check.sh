#!/bin/bash
# round 5, final visit: full GPU suite + variant tests + smoke + contract bench + de-mixed rocprof + the other bench lines, final code, one box.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r5final
mkdir -p "$OUT"
python -c "
import ctypes, os
for p in ('fastvideo_amd/libfvk_amd.so', 'scripts/probes/libfvk_probe.so'):
    ctypes.CDLL(os.path.abspath(p)); print('loads', p)
" || exit 1
( time timeout 1500 python -m pytest tests -m gpu -q -rs ) > "$OUT/pytest_full.log" 2>&1; echo "pytest rc=$?"; tail -6 "$OUT/pytest_full.log" | cut -c1-300
FVK_PROBE_LIB=1 timeout 600 python -m pytest scripts/probes/variant_tests.py -q > "$OUT/pytest_variants.log" 2>&1; echo "variants rc=$?"; tail -2 "$OUT/pytest_variants.log" | cut -c1-300
timeout 300 python -c "import __graft_entry__ as G; G.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > "$OUT/bench.log" 2> "$OUT/bench.err"; echo "bench rc=$?"; tail -1 "$OUT/bench.log" | cut -c1-2500
bash scripts/prof.sh r5final --no-cfg-step --no-vae --no-power-trace 2>&1 | grep -v "distribution\|at::native" | tail -16 | cut -c1-200
run() { name=$1; shift; timeout 400 python bench.py "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; echo "$name rc=$? $(python -c "
import json,sys
try:
    j=json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1]); r=j['roofline']
    print('ms/step', j['ms_per_step'], 'step TF', j.get('step_tflops'), '| roofline', r['achieved'], r['frac'], '| sclk', ((j.get('power') or {}).get('sclk_mhz') or {}).get('p50'))
except Exception as e: print('parse error', e)
")"; }
run sta --attention sta --no-cpu-baseline --no-vae --no-cfg-step
run vsa --attention vsa --no-cpu-baseline --no-vae --no-cfg-step
run fp8c --quant fp8_channel --no-cpu-baseline --no-vae --no-cfg-step
run cfg1 --config cfg1 --no-cpu-baseline --no-cfg-step --no-vae
