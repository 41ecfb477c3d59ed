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
3)
  # full GPU suite with the new block-sparse kernel, then its counters, then the VSA bench line
  timeout 1700 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
  timeout 300 python scripts/vsa_bs16_ab.py > $OUT/vsa_bs16_ab.log 2>&1; echo "bs16_ab rc=$?"
  i=0
  for SET in "GRBM_GUI_ACTIVE" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
             "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    PMC=1 ATTN_IMPL=0 N_LAUNCH=3 timeout 150 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/p$i" -o pmc -- python scripts/vsa_bs16_ab.py > "$OUT/p$i.log" 2>&1 < /dev/null
    echo "pass $i ($SET) rc=$? $(tail -1 $OUT/p$i.log | cut -c1-160)"
  done
  python - <<'PY'
import csv, glob, collections, json
ctr = collections.defaultdict(list); dur = []
for f in glob.glob("gpurun_out/r6v3/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_bs16_kernel" in r["Kernel_Name"]: ctr[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/r6v3/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_bs16_kernel" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
m = {k: sum(v[-3:]) / len(v[-3:]) for k, v in ctr.items()}   # the last three launches: the measured lists (the model's own forward launched it twice before)
d3 = dur[-3:]
res = dict(kernel="attn_bs16_kernel (block-sparse, one wave per 64-row list), cfg2 VSA lists of the model's second layer: 624 blocks, top-125, 12 heads",
           ms_under_profiler=round(sum(d3) / max(len(d3), 1), 4), **m)
if "GRBM_GUI_ACTIVE" in m and d3:
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    res["effective_clock_ghz"] = round(cyc / (res["ms_under_profiler"] * 1e-3) / 1e9, 3)
    for name, c, div in (("mfma_busy_fraction", "SQ_VALU_MFMA_BUSY_CYCLES", 1024), ("lds_active_fraction", "SQ_LDS_IDX_ACTIVE", 256)):
        if c in m: res[name] = round(m[c] / div / cyc, 3)
if "SQ_INSTS_LDS" in m and "SQ_INSTS_MFMA" in m:
    res["lds_instructions_per_mfma"] = round(m["SQ_INSTS_LDS"] / m["SQ_INSTS_MFMA"], 3)
    res["valu_instructions_per_mfma"] = round(m["SQ_INSTS_VALU"] / m["SQ_INSTS_MFMA"], 3)
if "SQ_WAIT_ANY" in m and "SQ_WAVE_CYCLES" in m: res["wait_any_over_wave_cycles"] = round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 3)
if "TCC_HIT_sum" in m: res["l2_hit_rate"] = round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 4)
if "FETCH_SIZE" in m: res["fabric_read_GB_per_launch"] = round(2 * m["FETCH_SIZE"] * 1024 / 1e9, 3)
json.dump(res, open("gpurun_out/r6v3/pmc_vsa_bs16.json", "w"), indent=1); print(json.dumps(res, indent=1))
PY
  find $OUT -name "*.csv" -size +2M -delete
  timeout 600 python bench.py --attention vsa --steps 10 --warmup 3 --no-cpu-baseline --no-vae > $OUT/bench_vsa.log 2>&1; echo "bench vsa rc=$?"; tail -1 $OUT/bench_vsa.log | cut -c1-1500
  ;;
4)
  # HIP-graph capture tests, host issue eager vs graphs (SP = 1 and 8 ranks sharing the GPU), store-policy A/B of the contract step
  timeout 900 python -m pytest tests/test_gpu_graph.py -x -q > $OUT/graph_tests.log 2>&1; echo "graph tests rc=$?"; tail -8 $OUT/graph_tests.log
  timeout 1700 python -m pytest tests -m gpu -q --deselect tests/test_gpu_graph.py > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
  timeout 1200 python scripts/graph_host_issue.py 8 > $OUT/host_issue.log 2>&1; echo "host issue rc=$?"; tail -40 $OUT/host_issue.log
  timeout 900 python scripts/step_tunable_ab.py '[["shipped (N >= 4096 streamed)", {}], ["N >= 3072 streamed (+ q|k)", {"gemm_impl": 3}], ["all streamed", {"gemm_impl": 4}], ["none streamed", {"gemm_impl": 6}], ["vt plain", {"gemm_impl": 2}]]' 4 > $OUT/store_ab.log 2>&1; echo "store_ab rc=$?"; grep forward_ms $OUT/store_ab.log | cut -c1-900
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vsa_prof -o vsa -- python bench.py --attention vsa --steps 3 --warmup 1 --no-cpu-baseline --no-vae --no-power-trace --no-matrix-ceiling > $OUT/vsa_prof.log 2>&1
  F=$(find $OUT/vsa_prof -name '*kernel_stats.csv' | head -1); [ -n "$F" ] && python scripts/condense_prof.py $F $OUT/vsa_kernel_stats.csv && head -30 $OUT/vsa_kernel_stats.csv
  find $OUT -name "*.csv" -size +2M -delete
  timeout 900 python bench.py --config cfg5 --attention vsa --quant fp8 --steps 3 --warmup 1 --no-cpu-baseline --no-vae --no-power-trace > $OUT/bench_cfg5_vsa_fp8.log 2>&1; echo "cfg5 vsa fp8 rc=$?"; tail -1 $OUT/bench_cfg5_vsa_fp8.log | cut -c1-700
  ;;
5)
  # guard-page allocator: every test that reaches the sparse kernels / the SP paths, each tensor ending at an unmapped page
  timeout 300 python scripts/guard_selftest.py > $OUT/guard_selftest.log 2>&1; echo "guard selftest rc=$? $(tail -1 $OUT/guard_selftest.log)"
  timeout 120 python scripts/guard_selftest.py oob > $OUT/guard_selftest_oob.log 2>&1; echo "guard oob rc=$? (non-zero + a fault message = the detector works) $(grep -i 'fault\|NO FAULT' $OUT/guard_selftest_oob.log | head -2)"
  for T in "tests/test_gpu_kernels.py" "tests/test_gpu_ref_triton.py" "tests/test_gpu_boundary.py" "tests/test_gpu_model.py" "tests/test_gpu_sp.py" \
           "tests/test_gpu_fullgeom.py::test_one_block_at_cfg2_vsa_matches_oracle tests/test_gpu_fullgeom.py::test_one_block_at_cfg2_sta_matches_oracle" \
           "tests/test_gpu_fullsize.py"; do
    N=$(echo $T | tr '/:. ' '____' | cut -c1-60)
    FVK_GUARD_ALLOC=1 timeout 1500 python -m pytest $T -x -q > $OUT/guard_$N.log 2>&1; echo "guard $T rc=$? $(tail -1 $OUT/guard_$N.log | cut -c1-150)"
    grep -i "memory access fault\|page not present\|guard_alloc\]" $OUT/guard_$N.log | sort | uniq -c | head -5
  done
  FVK_GUARD_ALLOC=1 FVK_GUARD_MODE=front timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ref_triton.py -x -q > $OUT/guard_front.log 2>&1; echo "guard front rc=$? $(tail -1 $OUT/guard_front.log | cut -c1-150)"
  ;;
7)
  timeout 400 python scripts/attn_lsum_ab.py > $OUT/attn_lsum_ab.log 2>&1; echo "lsum_ab rc=$?"; tail -30 $OUT/attn_lsum_ab.log
  timeout 900 python scripts/step_tunable_ab.py '[["attn_w16, row sums on the matrix pipe", {"attn_impl": 300}], ["attn_w16, row sums as VALU adds", {"attn_impl": 320}]]' 5 > $OUT/lsum_step_ab.log 2>&1; echo "lsum step rc=$?"; grep forward_ms $OUT/lsum_step_ab.log | cut -c1-700
  ;;
6)
  timeout 600 python scripts/vsa_bs16_ablate.py > $OUT/bs16_ablate.log 2>&1; echo "ablate rc=$?"; tail -30 $OUT/bs16_ablate.log
  timeout 1500 python -m pytest tests/test_gpu_bigseq.py -q --durations=0 > $OUT/bigseq_durations.log 2>&1; echo "bigseq rc=$?"; grep -E "passed|failed|s call|s setup" $OUT/bigseq_durations.log | head -20
  ;;
8)
  # which kernel faults under the guard-page allocator: every C-ABI call named and synchronised
  FVK_GUARD_ALLOC=1 FVK_TRACE_CALLS=1 timeout 600 python -m pytest "tests/test_gpu_model.py::test_wan_tiny_sta_matches_oracle" -x -q -s > $OUT/trace_sta.log 2>&1; echo "trace sta rc=$?"; grep "\[fvk\]\|fault" $OUT/trace_sta.log | tail -6
  FVK_GUARD_ALLOC=1 FVK_TRACE_CALLS=1 timeout 900 python -m pytest "tests/test_gpu_sp.py::test_sp_pipelined_exchange_equals_sp1" -x -q -s > $OUT/trace_sp.log 2>&1; echo "trace sp rc=$?"; grep "\[fvk\]\|fault" $OUT/trace_sp.log | tail -12
  ;;
9)
  # after the gate-row fix: the guard regression tests, then EVERY gpu test file under the guard-page allocator (one process per file)
  timeout 1500 python -m pytest tests/test_gpu_guard.py -x -q > $OUT/test_gpu_guard.log 2>&1; echo "test_gpu_guard rc=$? $(tail -1 $OUT/test_gpu_guard.log)"
  for T in tests/test_gpu_model.py tests/test_gpu_sp.py tests/test_gpu_fullgeom.py tests/test_gpu_fp8.py tests/test_gpu_sched.py tests/test_gpu_causal.py \
           tests/test_gpu_loader.py tests/test_gpu_reference_model.py tests/test_gpu_vae.py tests/test_gpu_vae_tiled.py tests/test_gpu_vae_real.py \
           tests/test_gpu_boundary.py tests/test_gpu_ref_triton.py tests/test_gpu_fullsize.py tests/test_gpu_rccl_single_rank.py; do
    [ -f $T ] || continue
    N=$(basename $T .py)
    FVK_GUARD_ALLOC=1 timeout 1500 python -m pytest $T -q > $OUT/guard_$N.log 2>&1; echo "guard $T rc=$? $(tail -1 $OUT/guard_$N.log | cut -c1-150)"
    grep -i "memory access fault" $OUT/guard_$N.log | sort | uniq -c | head -3
  done
  ;;
10)
  timeout 900 python scripts/sp_rank_emulation.py > $OUT/sp_rank_emulation.json 2> $OUT/sp_rank_emulation.err; echo "sp emulation rc=$?"; python -c "
import json
d=json.load(open('$OUT/sp_rank_emulation.json'))
for k,v in d.items(): print(k, v['layer_us'], v.get('layer_us_in_sequence_eager'), v.get('layer_us_in_sequence_hip_graph'), v.get('compute_only_speedup'), v.get('compute_only_speedup_in_sequence_eager'), v.get('compute_only_speedup_in_sequence_hip_graph'))
"
  timeout 1700 python -m pytest tests -m gpu -q --durations=25 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 $OUT/pytest_gpu.log
  ;;
11)
  timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ref_triton.py -x -q -k "sparse or vsa or block or triton" > $OUT/sparse_tests.log 2>&1; echo "sparse tests rc=$?"; tail -5 $OUT/sparse_tests.log
  timeout 300 python scripts/vsa_bs16_ab.py > $OUT/vsa_bs16_ab.log 2>&1; echo "bs16_ab rc=$?"; python -c "
import json
s=open('$OUT/vsa_bs16_ab.log').read(); j=json.loads(s[s.index('{'):])
for k,v in j.items(): print(k, {a:min(b) for a,b in v['ms'].items()}, v['bs16_vs_round1_max_abs'], v['max_abs_err_vs_exact_fp32_on_sampled_blocks'])
"
  ;;
esac
echo "visit $V done"
