#!/bin/bash
# Round 6 GPU visits (one gpurun call each): `bash scripts/round6_visit.sh <n>`; logs under gpurun_out/r6v<n>/, copied into profiles/ by hand.
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=${1:-1}; OUT=gpurun_out/r6v$V; mkdir -p $OUT
bash scripts/box_info.sh > $OUT/box_info.log 2>&1
case $V in
1)
  timeout 1500 python -m pytest tests/test_gpu_bigseq.py -x -q -s > $OUT/bigseq.log 2>&1; echo "bigseq rc=$?"; tail -5 $OUT/bigseq.log
  timeout 300 python scripts/vsa_xcd_ab.py > $OUT/vsa_xcd_ab.log 2>&1; echo "xcd_ab rc=$?"; tail -25 $OUT/vsa_xcd_ab.log
  for IMPL in 0 2; do
    PMC=1 VSA_IMPL=$IMPL N_LAUNCH=3 timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc$IMPL -o pmc -- python scripts/vsa_xcd_ab.py > $OUT/tcc$IMPL.log 2>&1
    PMC=1 VSA_IMPL=$IMPL N_LAUNCH=3 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch$IMPL -o pmc -- python scripts/vsa_xcd_ab.py > $OUT/fetch$IMPL.log 2>&1
  done
  python - <<'PY'
import csv, glob, collections, json
res = {}
for impl in (0, 2):
    ctr = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r6v1/tcc{impl}/**/*counter_collection.csv", recursive=True) + glob.glob(f"gpurun_out/r6v1/fetch{impl}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_fwd_kernel" in r["Kernel_Name"]: ctr[r["Counter_Name"]].append(float(r["Counter_Value"]))
    # the last 3 launches are the measured ones (the model's own forward launches the kernel twice before)
    m = {k: sum(v[-3:]) / len(v[-3:]) for k, v in ctr.items() if v}
    if "TCC_HIT_sum" in m: m["l2_hit_rate"] = round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 4)
    if "FETCH_SIZE" in m: m["fabric_read_GB_per_launch"] = round(2 * m["FETCH_SIZE"] * 1024 / 1e9, 3)
    res["round_robin" if impl == 0 else "xcd_contiguous"] = m
json.dump(res, open("gpurun_out/r6v1/vsa_xcd_pmc.json", "w"), indent=1); print(json.dumps(res, indent=1))
PY
  find $OUT -name "*.csv" -size +2M -delete
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vt_ab -o vt -- python scripts/step_tunable_ab.py '[["vt streaming stores (shipped)", {}], ["vt plain stores", {"gemm_impl": 2}]]' 4 > $OUT/vt_ab.log 2>&1; echo "vt_ab rc=$?"; tail -3 $OUT/vt_ab.log
  F=$(find $OUT/vt_ab -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && python scripts/condense_prof.py $F $OUT/vt_ab_kernel_stats.csv
  find $OUT -name "*.csv" -size +2M -delete
  ;;
2)
  # the new block-sparse kernel (attn_bs16.hip): every test that reaches it, then the kernel-level A/B against the round-1 kernel
  timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ref_triton.py tests/test_gpu_boundary.py -x -q -k "sparse or vsa or block or triton" > $OUT/sparse_tests.log 2>&1; echo "sparse tests rc=$?"; tail -15 $OUT/sparse_tests.log
  timeout 300 python scripts/vsa_bs16_ab.py > $OUT/vsa_bs16_ab.log 2>&1; echo "bs16_ab rc=$?"; tail -40 $OUT/vsa_bs16_ab.log
  ;;
esac
echo "visit $V done"
