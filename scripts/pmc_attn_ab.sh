#!/bin/bash
# PMC passes of two attention kernels on the same box (separate rocprofv3 runs, kernel-trace + --pmc only).  usage: scripts/pmc_attn_ab.sh <tag> <impl...>
set -u
TAG=${1:-r3}; shift
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for IMPL in "$@"; do
  OUT=gpurun_out/pmc/${TAG}_impl$IMPL
  mkdir -p "$OUT"
  i=0
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    ATTN_IMPL=$IMPL timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/p$i" -o pmc -- python scripts/attn_pmc_one.py > "$OUT/p$i.log" 2>&1 < /dev/null
    echo "impl $IMPL pass $i rc=$?"
  done
  python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
dur = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn_" not in k: continue
        k = k.replace("(anonymous namespace)::", "").replace("void ", "")[:40].replace(",", ";")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for f in glob.glob(out + "/p*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn_" not in k: continue
        k = k.replace("(anonymous namespace)::", "").replace("void ", "")[:40].replace(",", ";")
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
with open(out + "/summary.csv", "w") as fo:
    fo.write("kernel,counter,per_launch\n")
    for k in agg:
        d = sorted(dur[k]); line = f"{k},duration_ms_median_under_profiler,{d[len(d)//2]:.4f}"
        print(line); fo.write(line + "\n")
        for c, v in sorted(agg[k].items()):
            line = f"{k},{c},{v/cnt[(k,c)]:.6g}"
            print(line); fo.write(line + "\n")
PY
  find "$OUT" -name "*.csv" -size +2M -delete
done
