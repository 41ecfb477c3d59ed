#!/bin/bash
# round 4, last visit: full GPU suite + variant tests + smoke + contract bench + VSA line + rocprof kernel stats with the final code
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4final
mkdir -p "$OUT"
python -c "
import ctypes, os
for p in ('fastvideo_amd/libfvk_amd.so', 'scripts/probes/libfvk_probe.so'):
    ctypes.CDLL(os.path.abspath(p)); print('loads', p)
" || exit 1
( time timeout 2400 python -m pytest tests -m gpu -q -rs ) > "$OUT/pytest_full.log" 2>&1; echo "pytest rc=$?"; tail -6 "$OUT/pytest_full.log" | cut -c1-300
FVK_PROBE_LIB=1 timeout 900 python -m pytest scripts/probes/variant_tests.py -q > "$OUT/pytest_variants.log" 2>&1; echo "variants rc=$?"; tail -2 "$OUT/pytest_variants.log" | cut -c1-300
timeout 300 python -c "import __graft_entry__ as G; G.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > "$OUT/bench.log" 2> "$OUT/bench.err"; echo "bench rc=$?"; tail -1 "$OUT/bench.log" | cut -c1-3000
timeout 400 python bench.py --attention vsa --no-vae --no-cpu-baseline --no-cfg-step > "$OUT/bench_vsa.log" 2> "$OUT/bench_vsa.err"; echo "vsa rc=$?"; tail -1 "$OUT/bench_vsa.log" | cut -c1-700
bash scripts/prof.sh r4final 2>&1 | grep -v "distribution\|at::native" | tail -14 | cut -c1-200
for rep in 1 2 3 4 5; do   # the two-rank shared-GPU bench with the pipelined exchange: must keep "pipelined" every time
  FVK_BENCH_SHARED_GPU=1 FVK_SP_OVERLAP=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600+rep)) bench.py --gpus 2 --steps 2 --warmup 1 --layers 2 > "$OUT/b2_$rep.log" 2> "$OUT/b2_$rep.err"
  echo "two-rank rep $rep rc=$? $(tail -1 "$OUT/b2_$rep.log" | python -c "import sys,json; print(json.loads(sys.stdin.read())['config']['parallelism'][-60:])")  $(grep -o "rank [0-9]: [^;]*differ[^;]*" "$OUT/b2_$rep.err" | head -2 | tr '\n' ' ')"
done
