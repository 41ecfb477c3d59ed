#!/bin/bash
# round 4: PMC passes of the 3x3x3 conv kernels — vae_conv3w (IMPL 0: one wave per SIMD, 16x16x32 MFMAs, persistent) against vae_conv3 (IMPL 3: the
# 8-wave kernel) at the 96 -> 96 full-resolution and 192 -> 192 shapes.  One counter set per rocprofv3 run (--kernel-trace only), T = 8 frames per launch.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/conv_pmc_r4; mkdir -p $OUT
i=0
for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  for IMPL in 0 3; do for C in 96 192; do
    if [ $i -ge 4 ] && [ $C -ne 96 ]; then continue; fi
    IMPL=$IMPL C=$C N_LAUNCH=2 timeout 90 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/p${i}_${IMPL}_$C" -o pmc -- python scripts/conv96_only.py > "$OUT/p${i}_${IMPL}_$C.log" 2>&1 < /dev/null
    rc=$?; echo "pass $i impl $IMPL C=$C rc=$rc"; [ $rc -ne 0 ] && { grep -m1 -i "fault\|error" "$OUT/p${i}_${IMPL}_$C.log"; exit 1; }
  done; done
done
python - <<'PY'
import csv, glob, collections, json
res = {}
for impl in (0, 3):
    for C in (96, 192):
        ctr = collections.defaultdict(list); dur = []
        for f in glob.glob(f"gpurun_out/conv_pmc_r4/p*_{impl}_{C}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "vae_conv3" in r["Kernel_Name"]: ctr[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(f"gpurun_out/conv_pmc_r4/p3_{impl}_{C}/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "vae_conv3" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
        m = {k: sum(v) / len(v) for k, v in ctr.items()}
        m["launch_ms_under_profiler"] = sum(dur) / max(len(dur), 1)
        res[f"impl{impl}_C{C}"] = m
json.dump(res, open("gpurun_out/conv_pmc_r4/summary.json", "w"), indent=1)
for k, m in res.items():
    T, H, W = 8, (480 if "C96" in k else 240), (832 if "C96" in k else 416)
    C = 96 if "C96" in k else 192
    mfma_instr = 2.0 * T * H * W * C * 27 * C / (2 * 16 * 16 * 32 if k.startswith("impl0") else 2 * 32 * 32 * 16) / 64  # wave-level MFMA instructions... per lane group
    print(k, {a: (round(b, 1) if b < 1e6 else f"{b:.4g}") for a, b in m.items()})
PY
find $OUT -name "*.csv" -size +2M -delete
