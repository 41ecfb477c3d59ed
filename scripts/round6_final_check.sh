#!/bin/bash
# round 6, final visit: counter passes of the contract kernel for the final sources, full GPU suite + variant tests + smoke, the contract bench as the
# driver runs it, its de-mixed rocprof summary, and every other bench line — final code, one box.
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r6final
mkdir -p "$OUT"
bash scripts/box_info.sh > $OUT/box_info.log 2>&1
python -c "
import ctypes, os
for p in ('fastvideo_amd/libfvk_amd.so', 'scripts/probes/libfvk_probe.so', 'scripts/probes/libguard_alloc.so'):
    ctypes.CDLL(os.path.abspath(p)); print('loads', p)
" || exit 1
KERNEL=attn_w16 bash scripts/pmc_traffic.sh r6final_w16 2>&1 | tail -4
( time timeout 1700 python -m pytest tests -m gpu -q -rs ) > "$OUT/pytest_full.log" 2>&1; echo "pytest rc=$?"; tail -6 "$OUT/pytest_full.log" | cut -c1-300
FVK_PROBE_LIB=1 timeout 600 python -m pytest scripts/probes/variant_tests.py -q > "$OUT/pytest_variants.log" 2>&1; echo "variants rc=$?"; tail -2 "$OUT/pytest_variants.log" | cut -c1-300
timeout 300 python -c "import __graft_entry__ as G; G.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.log" 2> "$OUT/bench.err"; echo "bench rc=$?"; tail -1 "$OUT/bench.log" | cut -c1-3000
bash scripts/prof.sh r6final --no-cfg-step --no-vae --no-power-trace --no-matrix-ceiling 2>&1 | grep -v "distribution\|at::native" | tail -16 | cut -c1-200
run() { name=$1; shift; timeout 900 python bench.py "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; echo "$name rc=$? $(python -c "
import json,sys
try:
    j=json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1]); r=j['roofline']
    print('ms/step', j['ms_per_step'], 'step TF', j.get('step_tflops'), '| roofline', r['achieved'], r['frac'], '| ceiling', r.get('sustained_matrix_rate_at_cap_tf'), '| vae', (j.get('vae') or {}).get('ms_per_step'), '| sclk', ((j.get('power') or {}).get('sclk_mhz') or {}).get('p50'))
except Exception as e: print('parse error', e)
")"; }
run vsa --attention vsa --no-cpu-baseline --no-vae --no-cfg-step
run sta --attention sta --no-cpu-baseline --no-vae --no-cfg-step
run fp8 --quant fp8 --no-cpu-baseline --no-vae --no-cfg-step
run fp8c --quant fp8_channel --no-cpu-baseline --no-vae --no-cfg-step
run cfg1 --config cfg1 --no-cpu-baseline --no-cfg-step --no-vae
run cfg5 --config cfg5 --no-cpu-baseline --no-cfg-step --steps 3 --warmup 1
run cfg5_vsa_fp8 --config cfg5 --attention vsa --quant fp8 --no-cpu-baseline --no-vae --no-cfg-step --steps 3 --warmup 1
run cfg4 --config cfg4 --no-cpu-baseline --no-vae --no-cfg-step --steps 2 --warmup 1
echo "final check done"
