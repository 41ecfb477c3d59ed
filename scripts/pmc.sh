#!/bin/bash
# PMC passes (separate runs, kernel-trace only — never combined with other trace domains) for one kernel family; prints
# per-kernel counter sums.   usage: scripts/pmc.sh <tag> <kernel-substring> <python script + args...>
set -u
TAG=${1:-r1}; KSUB=${2:-attn_pp}; shift 2
CMD=${*:-scripts/attn_only.py}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc/$TAG
mkdir -p "$OUT"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/p$i" -o pmc -- python $CMD > "$OUT/p$i.log" 2>&1 < /dev/null
  echo "pass $i ($SET) rc=$?"
done
python - "$OUT" "$KSUB" <<'PY'
import csv, glob, sys, collections
out, ksub = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if ksub not in k: continue
        k = k.replace("(anonymous namespace)::", "").replace("void ", "")[:48].replace(",", ";")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
with open(out + "/summary.csv", "w") as fo:
    fo.write("kernel,counter,sum_over_launches,launches,per_launch\n")
    for k in agg:
        for c, v in sorted(agg[k].items()):
            n = cnt[(k, c)]
            line = f"{k},{c},{v:.6g},{n},{v/n:.6g}"
            print(line); fo.write(line + "\n")
PY
find "$OUT" -name "*.csv" -size +5M -delete
