#!/bin/bash
# round 5, fourth visit (tight timeouts): synthetic victims beside the real aggressors (fixed scalar variant), the real-kernel matrix against
# gemm_w1 built with BUILTIN MFMAs (libfvk_bug2.so), the bench tests (the two-rank flow with the power repeat on every rank).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r5v4
mkdir -p "$OUT"
timeout 420 python scripts/coresidency_victims.py 60 > "$OUT/coresidency_victims.log" 2>&1; echo "victims rc=$?"; grep -v "^W\|amdgpu.ids" "$OUT/coresidency_victims.log" | python -c "
import sys, json
for ln in sys.stdin:
    try: j = json.loads(ln)
    except Exception: print(ln.strip()[:200]); continue
    if 'wrong_launches' in j: print(f\"{j['wrong_launches']:>6} of {j['of']} launches wrong | {j['aggressor'][:60]:60s} | {j['victim']}\")
    else: print(f\"{j['wrong_results']:>10} wrong (lo {j['wrong_low_half']}, hi {j['wrong_high_half']}, loads {j['wrong_loaded_values']}) | {j['aggressor'][:60]:60s} | {j['victim']}\")
"
FVK_PROBE_LIB=bug2 timeout 300 python scripts/coresidency_matrix.py 60 > "$OUT/coresidency_matrix_builtin_mfma.log" 2>&1; echo "matrix(bug2) rc=$?"; grep -v '"wrong": 0,' "$OUT/coresidency_matrix_builtin_mfma.log" | grep -v "^W\|amdgpu.ids" | cut -c1-400 | tail -12
( time timeout 900 python -m pytest tests/test_gpu_bench.py -m gpu -q -x ) > "$OUT/pytest_bench.log" 2>&1; echo "pytest bench rc=$?"; tail -5 "$OUT/pytest_bench.log" | cut -c1-300
