#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3b
mkdir -p "$OUT"
echo "== w64 check"
timeout 600 python scripts/attn_w64_check.py > "$OUT/w64_check.log" 2>&1; echo "rc=$?"; tail -25 "$OUT/w64_check.log"
echo "== sched test"
timeout 300 python -m pytest tests/test_gpu_sched.py -q -x 2>&1 | tail -3
echo "== ablations (correct mapping: 120 half LDS reads, 121 no exp, 123 no DMA, 124 half LDS + no DMA)"
AB_ROUNDS=5 timeout 300 python scripts/attn_ab.py 0 120 123 124 > "$OUT/attn_ablate2.log" 2>&1; echo "rc=$?"; tail -6 "$OUT/attn_ablate2.log"
