#!/bin/bash
# One GPU-box visit: parity tests, kernel A/B microbench, contract bench, rocprof kernel stats.  Everything lands in gpurun_out/<tag>/.
# usage: scripts/history/gpu_round.sh <tag> [skip-list e.g. "pmc"]
set -u
TAG=${1:-r1}
SKIP=${2:-}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
echo "== tests"; timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.log"
if [[ "$SKIP" != *micro* ]]; then
  echo "== microbench"; timeout 400 python scripts/microbench.py > "$OUT/microbench.log" 2>&1; echo "microbench rc=$?"; cat "$OUT/microbench.log" | cut -c1-600
  cp gpurun_out/microbench.json "$OUT/" 2>/dev/null
fi
echo "== bench"; timeout 600 python bench.py > "$OUT/bench.log" 2>&1; echo "bench rc=$?"; tail -2 "$OUT/bench.log" | cut -c1-1500
if [[ "$SKIP" != *prof* ]]; then
  echo "== rocprof"; bash scripts/prof.sh "$TAG" 2>&1 | tail -32
fi
if [[ "$SKIP" != *pmc* ]]; then
  echo "== pmc"; bash scripts/pmc.sh "$TAG" 2>&1 | tail -60
fi
