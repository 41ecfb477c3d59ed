#!/bin/bash
# visit 12: split last round of attn_bs16 — tests, A/B at cfg2 and cfg5 lists, VSA bench line
export TMPDIR=/tmp
O=gpurun_out/r6v12; mkdir -p $O
bash scripts/box_info.sh > $O/box_info.log 2>&1
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "block_sparse or video_sparse" > $O/sparse_tests.log 2>&1; echo "sparse tests rc=$?"; tail -3 $O/sparse_tests.log
timeout 600 python -m pytest tests/test_gpu_ref_triton.py tests/test_gpu_graph.py -m gpu -x -q > $O/triton_graph_tests.log 2>&1; echo "triton/graph tests rc=$?"; tail -3 $O/triton_graph_tests.log
timeout 600 python scripts/vsa_bs16_ab.py > $O/vsa_bs16_ab.log 2>&1; echo "bs16_ab rc=$?"
python - <<'P'
import json,re
t=open('gpurun_out/r6v12/vsa_bs16_ab.log').read()
j=json.loads(t[t.index('{'):])
for k,v in j.items(): print(k,{a:min(b) for a,b in v['ms'].items()},v['split_vs_whole_max_abs'],v['split_vs_whole_rows_changed'],v['max_abs_err_vs_exact_fp32_on_sampled_blocks'])
P
GRID=cfg5 timeout 900 python scripts/vsa_bs16_ab.py > $O/vsa_bs16_ab_cfg5.log 2>&1; echo "bs16_ab cfg5 rc=$?"; tail -c 1500 $O/vsa_bs16_ab_cfg5.log | grep -A12 '"ms"' | head -30
timeout 900 python bench.py --attention vsa --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-cfg-step > $O/bench_vsa.json 2> $O/bench_vsa.err; echo "bench vsa rc=$?"; python -c "
import json;d=json.loads(open('$O/bench_vsa.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['roofline'].get('frac'))"
echo "visit 12 done"
