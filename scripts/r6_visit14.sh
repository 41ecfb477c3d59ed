#!/bin/bash
# visit 14: VSA combine fused into the sparse kernel's store — tests, VSA bench line + kernel stats
export TMPDIR=/tmp
O=gpurun_out/r6v14; mkdir -p $O
bash scripts/box_info.sh > $O/box_info.log 2>&1
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "sparse or vsa" > $O/sparse_tests.log 2>&1; echo "sparse tests rc=$?"; tail -5 $O/sparse_tests.log
timeout 900 python -m pytest tests/test_gpu_ref_triton.py tests/test_gpu_graph.py tests/test_gpu_boundary.py -m gpu -x -q > $O/tests2.log 2>&1; echo "triton/graph/boundary tests rc=$?"; tail -3 $O/tests2.log
timeout 1200 python -m pytest tests/test_gpu_fullgeom.py tests/test_gpu_sp.py -m gpu -x -q -k "vsa or sparse" > $O/tests3.log 2>&1; echo "fullgeom/sp vsa tests rc=$?"; tail -3 $O/tests3.log
for i in 1 2; do
timeout 900 python bench.py --attention vsa --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-cfg-step > $O/bench_vsa_$i.json 2> $O/bench_vsa_$i.err; echo "bench vsa rc=$?"; python -c "
import json;d=json.loads(open('$O/bench_vsa_$i.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['roofline'].get('frac'))"
done
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_vsa -o vsa -- python $GRAFT_REPO_ROOT/bench.py --attention vsa --steps 3 --warmup 1 --no-cpu-baseline --no-vae --no-cfg-step --no-power-trace --no-matrix-ceiling > /tmp/prof_vsa.log 2>&1; cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_vsa -name "*kernel_stats.csv" | head -1); cp "$f" $O/vsa_kernel_stats.csv 2>/dev/null; head -22 $O/vsa_kernel_stats.csv | cut -c1-150
echo "visit 14 done"
