#!/bin/bash
# round 4, last visit: full GPU suite + variant tests + smoke + contract bench + VSA line + rocprof kernel stats with the final code
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4final
mkdir -p "$OUT"
python -c "
import ctypes, os
for p in ('fastvideo_amd/libfvk_amd.so', 'scripts/probes/libfvk_probe.so'):
    ctypes.CDLL(os.path.abspath(p)); print('loads', p)
" || exit 1
( time timeout 2400 python -m pytest tests -m gpu -q -rs ) > "$OUT/pytest_full.log" 2>&1; echo "pytest rc=$?"; tail -6 "$OUT/pytest_full.log" | cut -c1-300
FVK_PROBE_LIB=1 timeout 900 python -m pytest scripts/probes/variant_tests.py -q > "$OUT/pytest_variants.log" 2>&1; echo "variants rc=$?"; tail -2 "$OUT/pytest_variants.log" | cut -c1-300
timeout 300 python -c "import __graft_entry__ as G; G.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > "$OUT/bench.log" 2> "$OUT/bench.err"; echo "bench rc=$?"; tail -1 "$OUT/bench.log" | cut -c1-3000
timeout 400 python bench.py --attention vsa --no-vae --no-cpu-baseline --no-cfg-step > "$OUT/bench_vsa.log" 2> "$OUT/bench_vsa.err"; echo "vsa rc=$?"; tail -1 "$OUT/bench_vsa.log" | cut -c1-700
bash scripts/prof.sh r4final 2>&1 | grep -v "distribution\|at::native" | tail -14 | cut -c1-200
bash scripts/conv_pmc_traffic.sh r4final 2>&1 | grep -i "rc=\|traffic_over\|without_fetch"
KERNEL=attn_w16 bash scripts/pmc_traffic.sh r4final 2>&1 | grep -i "rc=\|traffic_bytes\|effective_clock\|mfma_busy"
