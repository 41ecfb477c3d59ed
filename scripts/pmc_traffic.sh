#!/bin/bash
# PMC passes for the dominant kernel of the contract bench (attn_w16_kernel at the cfg2 self-attention shape; KERNEL=attn_pp2 for the round-1/2 kernel), each in its OWN rocprofv3
# run with --kernel-trace only (never combined with other trace domains), as guides/MI355X_MICROARCH.md prescribes:
#   FETCH_SIZE | WRITE_SIZE | GRBM_GUI_ACTIVE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
# Writes gpurun_out/pmc/<tag>/pmc_<kernel>.json (copy to profiles/<round>_pmc_<kernel>.json) with the gfx950 FETCH correction (x2 for wide
# streaming reads), the effective clock (GRBM_GUI_ACTIVE per XCD / kernel duration) and the sha256 of the kernel source measured — bench.py
# only reports `roofline.traffic` from a pass whose hash matches the source in the tree.
# usage: scripts/pmc_traffic.sh <tag>            (KERNEL=attn_w16 | attn_w64 | attn_pp2, default attn_w16 = the kernel fvk_attn_dense_bf16 ships)
set -u
TAG=${1:-r3}
export KERNEL=${KERNEL:-attn_w16}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc/$TAG
mkdir -p "$OUT"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  ATTN_IMPL=$([ "$KERNEL" = attn_pp2 ] && echo 99 || ([ "$KERNEL" = attn_w64 ] && echo 200 || echo 0)) N_LAUNCH=3 timeout ${PASS_TIMEOUT:-120} rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/p$i" -o pmc -- python scripts/attn_pmc_one.py > "$OUT/p$i.log" 2>&1 < /dev/null
  rc=$?
  echo "pass $i ($SET) rc=$rc"
  # a pass that faults or hangs (seen once: "Memory access fault" inside rocprofv3's own start-up on a box whose GPU then hung every later
  # process) must not burn the remaining passes' time-outs
  if [ $rc -ne 0 ]; then echo "aborting the PMC passes after a failed pass: $(grep -m1 -i "fault\|error" "$OUT/p$i.log")"; break; fi
done
python - "$OUT" <<'PY'
import csv, glob, hashlib, json, os, sys, collections
out = sys.argv[1]
KERNEL = os.environ.get("KERNEL", "attn_w16")
ctr = collections.defaultdict(list)
dur = []
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            ctr[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/p*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
mean = lambda k: sum(ctr[k]) / len(ctr[k]) if ctr.get(k) else None
S, H, D = 32760, 12, 128
res = {"kernel": KERNEL + "_kernel", "shape": f"B=1 S={S} H={H} D={D} (cfg2 self-attention)",
       "source": "scripts/pmc_traffic.sh: rocprofv3 --kernel-trace --pmc <one counter set per pass> -- python scripts/attn_pmc_one.py",
       "kernel_source_sha256": hashlib.sha256(open(f"fastvideo_amd/csrc/{KERNEL}.hip", "rb").read()).hexdigest(),
       "launches_per_pass": len(ctr.get("FETCH_SIZE", [])), "mean_launch_ms_under_profiler": round(sum(dur) / len(dur), 4) if dur else None}
fs, ws = mean("FETCH_SIZE"), mean("WRITE_SIZE")
if fs is not None and ws is not None:
    res.update(FETCH_SIZE_KB_per_launch=fs, WRITE_SIZE_KB_per_launch=ws,
               gfx950_correction="FETCH_SIZE counts 128-B requests at 64 B for 16-B/lane streaming reads (MI355X_MICROARCH.md, HBM section): fetch bytes = 2 x FETCH_SIZE",
               traffic_bytes_per_launch=int(2 * fs * 1024 + ws * 1024),
               algorithmic_bytes_per_launch=4 * S * H * D * 2)
ga = mean("GRBM_GUI_ACTIVE")
if ga is not None and dur:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3's aggregation: per-XCD busy cycles / wall = effective shader clock
    res.update(GRBM_GUI_ACTIVE_sum=ga, effective_clock_ghz=round(ga / 8 / (sum(dur) / len(dur) * 1e-3) / 1e9, 3))
for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT"):
    if mean(k) is not None:
        res[k] = mean(k)
if res.get("SQ_VALU_MFMA_BUSY_CYCLES") and res.get("GRBM_GUI_ACTIVE_sum"):
    # MFMA busy cycles are summed over 1024 SIMDs; GUI_ACTIVE/8 = cycles the chip was busy
    res["mfma_busy_fraction"] = round(res["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (res["GRBM_GUI_ACTIVE_sum"] / 8), 3)
json.dump(res, open(out + f"/pmc_{KERNEL}.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find "$OUT" -name "*.csv" -size +5M -delete
