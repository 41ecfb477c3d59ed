#!/bin/bash
# round 5, third visit: synthetic victims beside the real aggressors (co-residency bug, step 3), a longer flags A/B, the FULL GPU suite.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r5v3
mkdir -p "$OUT"
timeout 900 python scripts/coresidency_victims.py 60 > "$OUT/coresidency_victims.log" 2>&1; echo "victims rc=$?"; grep -v "^W\|amdgpu.ids" "$OUT/coresidency_victims.log" | cut -c1-330
timeout 900 python scripts/step_flags_ab.py 12 > "$OUT/step_flags_ab.log" 2>&1; echo "flags ab rc=$?"; tail -1 "$OUT/step_flags_ab.log" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['median_ms'], j['bit_identical_to_shipped'])"
( time timeout 2400 python -m pytest tests -m gpu -q -rs ) > "$OUT/pytest_full.log" 2>&1; echo "pytest rc=$?"; tail -6 "$OUT/pytest_full.log" | cut -c1-300
FVK_PROBE_LIB=1 timeout 900 python -m pytest scripts/probes/variant_tests.py -q > "$OUT/pytest_variants.log" 2>&1; echo "variants rc=$?"; tail -2 "$OUT/pytest_variants.log" | cut -c1-300
