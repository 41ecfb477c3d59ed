#!/bin/bash
# rocprofv3 kernel-trace + stats of any python script; prints total kernel time and the top kernels.  usage: scripts/prof_any.sh <tag> <script> [args...]
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof/$TAG
mkdir -p "$OUT"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o p -- python "$@" > "$OUT/run.log" 2>&1 < /dev/null
echo "rc=$?"; tail -1 "$OUT/run.log" | cut -c1-600
python - "$OUT" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("total kernel ms", sum(float(r["TotalDurationNs"]) for r in rows) / 1e6, "launches", sum(int(r["Calls"]) for r in rows))
for r in rows[:16]:
    print("%9.2f ms %6d calls avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:90]))
PY
find "$OUT" -name "*kernel_trace.csv" -size +5M -delete
