#!/bin/bash
# round 3, last visit: full GPU suite + variant tests + contract bench + rocprof kernel stats with the final code
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3final
mkdir -p "$OUT"
timeout 1800 python -m pytest tests -m gpu -q > "$OUT/pytest_full.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest_full.log"
timeout 600 python bench.py > "$OUT/bench.log" 2> "$OUT/bench.err"; echo "bench rc=$?"; tail -1 "$OUT/bench.log" | cut -c1-1500
bash scripts/prof.sh r3final --no-vae 2>&1 | grep -v "distribution\|at::native" | tail -12 | cut -c1-200
python - <<'PY'
import __graft_entry__ as G
G.smoke(); print("smoke ok")
PY
