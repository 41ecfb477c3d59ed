#!/bin/bash
# round 5, fifth visit: which part of gemm_w1 is the aggressor of the co-residency bug (stripped builds), tight timeouts.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r5v5
mkdir -p "$OUT"
for L in bug bug_s1 bug_s2 bug_s4 bug_s7 bug_s8 bug_s9; do
  FVK_PROBE_LIB=$L timeout 150 python scripts/coresidency_strips.py 40 > "$OUT/strips_$L.log" 2>&1; echo "$L rc=$?"
  grep -v "^W\|amdgpu.ids" "$OUT/strips_$L.log" | cut -c1-420
done
cat "$OUT"/strips_*.log | grep "^{" > "$OUT/coresidency_strips.log"
