#!/bin/bash
# round 4, box visit 7: the WHOLE GPU suite with the round's code, the contract line under the power sampler, rocprof kernel stats, conv PMC passes,
# VSA union-walk overlap, SP rank emulation
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4g; mkdir -p $OUT
python -c "import ctypes,torch; ctypes.CDLL('fastvideo_amd/libfvk_amd.so'); ctypes.CDLL('scripts/probes/libfvk_probe.so'); print('both libraries load')" || exit 1
echo "== full suite"; timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_all.log 2>&1; echo rc=$?; tail -25 $OUT/pytest_all.log | cut -c1-400
echo "== bench"; timeout 600 python scripts/power_trace.py --out $OUT/bench_power -- python bench.py --steps 20 --warmup 5 > $OUT/bench.log 2> $OUT/bench.err; echo rc=$?; tail -1 $OUT/bench.log | cut -c1-300; tail -2 $OUT/bench.err | cut -c1-600
echo "== rocprof contract + vae"; bash scripts/prof.sh r4g 2>&1 | grep -v "distribution\|at::native" | tail -25 | cut -c1-220
echo "== vsa overlap"; timeout 200 python scripts/vsa_overlap.py > $OUT/vsa_overlap.log 2>&1; tail -1 $OUT/vsa_overlap.log | cut -c1-300
echo "== conv pmc"; bash scripts/conv_pmc_r4.sh 2>&1 | tail -12 | cut -c1-700
