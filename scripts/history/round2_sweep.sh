#!/bin/bash
# Round-2 measurement sweep on one GPU box (outputs under gpurun_out/<tag>/): the whole -m gpu suite, every bench line (contract line with the
# reference-kind CPU baseline, sparse / fp8 / cfg1 / cfg5 / cfg4 lines, VAE stage), and rocprofv3 --kernel-trace --stats summaries.
set -u
TAG=${1:-r2sweep}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
( time timeout 900 python -m pytest tests -q -m gpu -rs ) > "$OUT/pytest_all.log" 2>&1; echo "pytest rc=$? $(tail -4 "$OUT/pytest_all.log" | head -1)"
timeout 400 python bench.py --steps 20 --warmup 3 > "$OUT/bench_contract.log" 2> "$OUT/bench_contract.err"; echo "contract rc=$? $(tail -1 "$OUT/bench_contract.log" | cut -c1-600)"
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" > "$OUT/$name.log" 2>&1; echo "$name rc=$? $(tail -1 "$OUT/$name.log" | cut -c1-330)"; }
run cfg2_sta --attention sta --steps 5 --warmup 2
run cfg2_vsa --attention vsa --steps 5 --warmup 2
run cfg2_fp8 --quant fp8 --steps 5 --warmup 2
run cfg2_fp8_channel --quant fp8_channel --steps 5 --warmup 2
run cfg1 --config cfg1 --steps 5 --warmup 2
run cfg5_dense --config cfg5 --steps 2 --warmup 1
run cfg5_vsa_fp8 --config cfg5 --attention vsa --quant fp8 --steps 2 --warmup 1
run cfg4 --config cfg4 --steps 2 --warmup 1
timeout 400 python bench.py --stage vae --steps 3 --warmup 1 > "$OUT/vae_cfg2.log" 2> "$OUT/vae_cfg2.err"; echo "vae cfg2 rc=$? $(tail -1 "$OUT/vae_cfg2.log" | cut -c1-500)"
timeout 400 python bench.py --stage vae --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/vae_cfg5.log" 2>&1; echo "vae cfg5 rc=$? $(tail -1 "$OUT/vae_cfg5.log" | cut -c1-500)"
timeout 300 python scripts/causal_bench.py > "$OUT/causal_480p.log" 2>&1; echo "causal rc=$? $(tail -1 "$OUT/causal_480p.log" | cut -c1-400)"
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
prof() { name=$1; shift; timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$name" -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/prof_$name.log" 2>&1 < /dev/null; echo "prof $name rc=$?"; F=$(find "$OUT/prof_$name" -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && python scripts/condense_prof.py "$F" "$OUT/${name}_kernel_stats.csv"; find "$OUT/prof_$name" -name "*kernel_trace.csv" -delete; }
prof dense
prof vsa --attention vsa
prof sta --attention sta
prof vae --stage vae
