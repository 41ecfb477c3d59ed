#!/bin/bash
# round 5, sixth visit: the stand-alone two-kernel repro of the co-residency fault (no library code).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r5v6
mkdir -p "$OUT"
timeout 240 scripts/probes/coresidency_repro 30 > "$OUT/coresidency_repro.log" 2>&1; echo "repro rc=$?"
python - "$OUT/coresidency_repro.log" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    try: j = json.loads(ln)
    except Exception: print(ln.strip()[:200]); continue
    print(f"{j['wrong']:>10} wrong (lo {j['wrong_low_half']:>9}, hi {j['wrong_high_half']:>6}) | agg {j['aggressor_ms']:6.1f} ms vic {j['victim_ms']:6.1f} ms | {j['aggressor'][:78]:78s} | {j['victim'][:40]}")
PY
