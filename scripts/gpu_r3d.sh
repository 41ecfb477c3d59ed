#!/bin/bash
# round 3, GPU visit D: FULL gpu test suite on the product library (attn_w64 shipped), contract bench, PMC traffic pass, rocprof kernel stats
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3d
mkdir -p "$OUT"
echo "== full gpu suite"
timeout 1800 python -m pytest tests -m gpu -q --durations=10 > "$OUT/pytest_full.log" 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error" "$OUT/pytest_full.log" | tail -3; grep -E "^(FAILED|ERROR)" "$OUT/pytest_full.log" | head -20
echo "== pmc traffic (attn_w64)"
PASS_TIMEOUT=100 bash scripts/pmc_traffic.sh r3d > "$OUT/pmc_traffic.log" 2>&1; tail -25 "$OUT/pmc_traffic.log"
cp gpurun_out/pmc/r3d/pmc_attn_w64.json profiles/r03d_pmc_attn_w64.json 2>/dev/null
echo "== bench"
timeout 600 python bench.py > "$OUT/bench.log" 2> "$OUT/bench.err"; echo "bench rc=$?"; tail -1 "$OUT/bench.log" | cut -c1-1800
echo "== rocprof"
bash scripts/prof.sh r3d --no-vae 2>&1 | tail -22
