#!/bin/bash
# round 4, box visit 1: the new conv kernel (parity, A/B by conv shape), the whole GPU suite, the contract line under the power sampler
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4a; mkdir -p $OUT
python scripts/power_trace.py --out $OUT/idle_power --hz 20 -- sleep 2 > $OUT/idle.log 2>&1; tail -2 $OUT/idle.log | cut -c1-600
echo "== vae tests"; timeout 300 python -m pytest tests/test_gpu_vae.py -q -x > $OUT/vae_tests.log 2>&1; echo rc=$?; tail -15 $OUT/vae_tests.log | cut -c1-400
echo "== conv3w vs 8-wave"; FVK_PROBE_LIB=1 timeout 300 python -m pytest scripts/probes/variant_tests.py -q -k conv3w > $OUT/variant.log 2>&1; echo rc=$?; tail -12 $OUT/variant.log | cut -c1-400
echo "== breakdown new"; timeout 200 python scripts/vae_conv_breakdown.py > $OUT/vae_breakdown_new.log 2>&1; cat $OUT/vae_breakdown_new.log | cut -c1-200
echo "== breakdown old"; FVK_PROBE_LIB=1 timeout 200 python scripts/vae_conv_breakdown.py --impl 3 > $OUT/vae_breakdown_old.log 2>&1; cat $OUT/vae_breakdown_old.log | cut -c1-200
echo "== full suite"; timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_all.log 2>&1; echo rc=$?; tail -40 $OUT/pytest_all.log | cut -c1-300
echo "== bench"; timeout 600 python scripts/power_trace.py --out $OUT/bench_power -- python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2> $OUT/bench.err; echo rc=$?; tail -1 $OUT/bench.log | cut -c1-6000; tail -3 $OUT/bench.err | cut -c1-800
cp gpurun_out/full_contract_parity.json $OUT/ 2>/dev/null
