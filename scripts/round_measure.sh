#!/bin/bash
# Round-end measurement sweep on one GPU box: every BASELINE config's bench line + VAE decode + PMC passes of the GEMM kernel.
# usage: scripts/round_measure.sh <tag>     (outputs under gpurun_out/<tag>/)
set -u
TAG=${1:-meas}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" > "$OUT/$name.log" 2>&1; echo "$name rc=$? $(tail -1 "$OUT/$name.log" | cut -c1-330)"; }
run cfg2_dense --steps 5 --warmup 2
run cfg2_sta --attention sta --steps 5 --warmup 2
run cfg2_vsa --attention vsa --steps 5 --warmup 2
run cfg2_fp8 --quant fp8 --steps 5 --warmup 2
run cfg1 --config cfg1 --steps 5 --warmup 2
run cfg5_dense --config cfg5 --steps 2 --warmup 1
run cfg5_vsa_fp8 --config cfg5 --attention vsa --quant fp8 --steps 2 --warmup 1
run cfg4 --config cfg4 --steps 2 --warmup 1
timeout 300 python scripts/causal_bench.py > "$OUT/causal_480p.log" 2>&1; echo "causal rc=$? $(tail -1 "$OUT/causal_480p.log" | cut -c1-500)"
timeout 300 python scripts/vae_bench.py > "$OUT/vae_480p.log" 2>&1; echo "vae480 rc=$? $(tail -1 "$OUT/vae_480p.log" | cut -c1-300)"
timeout 300 python scripts/vae_bench.py --frames 33 --h 90 --w 160 > "$OUT/vae_720p.log" 2>&1; echo "vae720 rc=$? $(tail -1 "$OUT/vae_720p.log" | cut -c1-300)"
# PMC passes of the bf16 GEMM kernel at the FFN-out shape (separate runs, kernel-trace + pmc only)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/pmc/p$i" -o pmc -- scripts/probes/gemm_harness.bin 0 --only=ffn_out > "$OUT/pmc_p$i.log" 2>&1 < /dev/null
  echo "pmc pass $i rc=$?"
done
python - "$OUT/pmc" gemm_ph <<'PY'
import csv, glob, sys, collections
out, ksub = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if ksub not in k: continue
        k = k.replace("(anonymous namespace)::", "").replace("void ", "")[:48].replace(",", ";")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
with open(out + "/summary.csv", "w") as fo:
    fo.write("kernel,counter,sum_over_launches,launches,per_launch\n")
    for k in agg:
        for c, v in sorted(agg[k].items()):
            n = cnt[(k, c)]; line = f"{k},{c},{v:.6g},{n},{v/n:.6g}"; print(line); fo.write(line + "\n")
PY
find "$OUT" -name "*.csv" -size +5M -delete
