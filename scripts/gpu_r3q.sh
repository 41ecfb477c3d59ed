#!/bin/bash
# round 3, final measurement visit: contract bench, PMC traffic pass of the dominant kernel (attn_w16), rocprof kernel stats of the bench command
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3q
mkdir -p "$OUT"
echo "== pmc traffic (attn_w16)"
PASS_TIMEOUT=100 bash scripts/pmc_traffic.sh r3q > "$OUT/pmc_traffic.log" 2>&1; tail -12 "$OUT/pmc_traffic.log"
cp gpurun_out/pmc/r3q/pmc_attn_w16.json profiles/r03q_pmc_attn_w16.json 2>/dev/null
echo "== bench"
timeout 600 python bench.py > "$OUT/bench.log" 2> "$OUT/bench.err"; echo "bench rc=$?"; tail -1 "$OUT/bench.log" | cut -c1-2400
echo "== rocprof"
bash scripts/prof.sh r3q --no-vae 2>&1 | tail -24
