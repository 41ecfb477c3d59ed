#!/bin/bash
# PMC passes of the sliding-tile list kernel (attn_pp2_kernel<...,LIST>) at the cfg2 grid: one counter set per rocprofv3 run (kernel trace only).
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/sta_pmc; mkdir -p $OUT
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  N_LAUNCH=3 timeout 100 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/p$i" -o pmc -- python scripts/sta_only.py > "$OUT/p$i.log" 2>&1 < /dev/null
  rc=$?; echo "pass $i rc=$rc $(tail -1 $OUT/p$i.log | cut -c1-160)"; [ $rc -ne 0 ] && { grep -m1 -i "fault\|error" "$OUT/p$i.log"; exit 1; }
done
python - <<'PY'
import csv, glob, collections, json
ctr = collections.defaultdict(list); dur = []
for f in glob.glob("gpurun_out/sta_pmc/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_pp2" in r["Kernel_Name"]: ctr[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/sta_pmc/p3/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_pp2" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
m = {k: sum(v) / len(v) for k, v in ctr.items()}
ms = sum(dur) / len(dur)
res = dict(kernel="attn_pp2_kernel<...,LIST> (sliding-tile, queries packed by window class), grid 21x30x52, 12 heads", ms_under_profiler=round(ms, 4), **m)
res["traffic_bytes_per_launch"] = int(2 * m["FETCH_SIZE"] * 1024 + m["WRITE_SIZE"] * 1024)
res["effective_clock_ghz"] = round(m["GRBM_GUI_ACTIVE"] / 8 / (ms * 1e-3) / 1e9, 3)
res["mfma_busy_fraction"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (m["GRBM_GUI_ACTIVE"] / 8), 3)
res["lds_active_fraction"] = round(m["SQ_LDS_IDX_ACTIVE"] / 256 / (m["GRBM_GUI_ACTIVE"] / 8), 3)
json.dump(res, open("gpurun_out/sta_pmc/pmc_sta_lists.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find $OUT -name "*.csv" -size +2M -delete
