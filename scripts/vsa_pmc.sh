#!/bin/bash
# Counter passes of the block-sparse kernel attn_bs16 at the cfg2 VSA lists of the model's second layer (scripts/vsa_bs16_ab.py in PMC mode), one
# counter set per rocprofv3 run, kernel trace only.  Summary: gpurun_out/vsa_pmc/pmc_vsa_bs16.json (round 6: profiles/r06b_pmc_vsa_bs16.json).
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/vsa_pmc; mkdir -p $OUT
i=0
for SET in "GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  PMC=1 ATTN_IMPL=${ATTN_IMPL:-0} N_LAUNCH=3 timeout 150 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/p$i" -o pmc -- python scripts/vsa_bs16_ab.py > "$OUT/p$i.log" 2>&1 < /dev/null
  echo "pass $i ($SET) rc=$?"
done
python - <<'PY'
import csv, glob, collections, json
ctr = collections.defaultdict(list); dur = []
for f in glob.glob("gpurun_out/vsa_pmc/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_bs16_kernel" in r["Kernel_Name"] or "attn_fwd_kernel" in r["Kernel_Name"]: ctr[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/vsa_pmc/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_bs16_kernel" in r["Kernel_Name"] or "attn_fwd_kernel" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
m = {k: sum(v[-3:]) / len(v[-3:]) for k, v in ctr.items()}   # the last three launches: the measured lists
d3 = dur[-3:]
res = dict(ms_under_profiler=round(sum(d3) / max(len(d3), 1), 4), **m)
if "GRBM_GUI_ACTIVE" in m and d3:
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    res["effective_clock_ghz"] = round(cyc / (res["ms_under_profiler"] * 1e-3) / 1e9, 3)
    for name, c, div in (("mfma_busy_fraction", "SQ_VALU_MFMA_BUSY_CYCLES", 1024), ("lds_active_fraction", "SQ_LDS_IDX_ACTIVE", 256)):
        if c in m: res[name] = round(m[c] / div / cyc, 3)
if "SQ_INSTS_VALU" in m and "SQ_INSTS_MFMA" in m: res["valu_instructions_per_mfma"] = round(m["SQ_INSTS_VALU"] / m["SQ_INSTS_MFMA"], 3)
if "SQ_WAIT_ANY" in m and "SQ_WAVE_CYCLES" in m: res["wait_any_over_wave_cycles"] = round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 3)
if "TCC_HIT_sum" in m: res["l2_hit_rate"] = round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 4)
if "FETCH_SIZE" in m: res["fabric_read_GB_per_launch"] = round(2 * m["FETCH_SIZE"] * 1024 / 1e9, 3)
json.dump(res, open("gpurun_out/vsa_pmc/pmc_vsa_bs16.json", "w"), indent=1); print(json.dumps(res, indent=1))
PY
find $OUT -name "*.csv" -size +2M -delete
