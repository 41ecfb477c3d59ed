#!/bin/bash
# Sparse-attention measurement pass (outputs under gpurun_out/<tag>/): STA / VSA bench lines, the STA list-form A/B, rocprofv3 kernel stats.
set -u
TAG=${1:-sparse}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" > "$OUT/$name.log" 2>&1; echo "$name rc=$? $(tail -1 "$OUT/$name.log" | cut -c1-330)"; }
run cfg2_sta --attention sta --steps 10 --warmup 2
run cfg2_vsa --attention vsa --steps 10 --warmup 2
run cfg5_vsa_fp8 --config cfg5 --attention vsa --quant fp8 --steps 2 --warmup 1
run cfg5_sta --config cfg5 --attention sta --steps 2 --warmup 1
timeout 200 python scripts/sta_lists_ab.py > "$OUT/sta_lists_ab_21x30x52.log" 2>&1; tail -3 "$OUT/sta_lists_ab_21x30x52.log"
timeout 200 python scripts/sta_lists_ab.py 18 48 80 > "$OUT/sta_lists_ab_18x48x80.log" 2>&1; tail -3 "$OUT/sta_lists_ab_18x48x80.log"
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
prof() { name=$1; shift; timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$name" -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/prof_$name.log" 2>&1 < /dev/null; echo "prof $name rc=$?"; F=$(find "$OUT/prof_$name" -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && python scripts/condense_prof.py "$F" "$OUT/${name}_kernel_stats.csv"; find "$OUT/prof_$name" -name "*kernel_trace.csv" -delete; }
prof sta --attention sta
prof vsa --attention vsa
