#!/bin/bash
# rocprofv3 kernel-trace + stats of the contract bench; summaries land in gpurun_out/prof/<tag>/ (copy to profiles/).
# usage: scripts/prof.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof/$TAG
mkdir -p "$OUT"
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench -- \
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1 < /dev/null
echo "rocprofv3 rc=$?"
tail -1 "$OUT/bench.log" | cut -c1-400
find "$OUT" -type f | head -20
F=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then head -30 "$F"; fi
# keep only the small summaries (the raw trace can be large)
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
