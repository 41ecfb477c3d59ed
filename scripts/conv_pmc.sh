cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/conv_pmc; mkdir -p $OUT
for C in 96 192; do C=$C N_LAUNCH=5 timeout 60 python scripts/conv96_only.py 2>&1 | tail -1; done
IMPL=2 N_LAUNCH=5 timeout 60 python scripts/conv96_only.py 2>&1 | tail -1
i=0
for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for C in 96 192; do
    C=$C N_LAUNCH=2 timeout 90 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/p${i}_$C" -o pmc -- python scripts/conv96_only.py > "$OUT/p${i}_$C.log" 2>&1 < /dev/null
    rc=$?; echo "pass $i C=$C rc=$rc"; [ $rc -ne 0 ] && { grep -m1 -i "fault\|error" "$OUT/p${i}_$C.log"; exit 1; }
  done
done
python - <<'PY'
import csv, glob, collections
for C in (96, 192):
    ctr = collections.defaultdict(list); dur=[]
    for f in glob.glob(f"gpurun_out/conv_pmc/p*_{C}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "vae_conv3" in r["Kernel_Name"]: ctr[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(f"gpurun_out/conv_pmc/p3_{C}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "vae_conv3" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    print(C, {k: sum(v)/len(v) for k, v in ctr.items()}, "ms", sum(dur)/max(len(dur),1))
PY
find $OUT -name "*.csv" -size +2M -delete
