#!/bin/bash
# round 4, box visit 2: gemm_w1n + pipelined SP + bench tests, conv3w tile-time probe, per-rank GEMM shapes, SP rank emulation
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4b; mkdir -p $OUT
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sp.py tests/test_gpu_bench.py tests/test_gpu_boundary.py -q > $OUT/tests.log 2>&1; echo rc=$?; tail -25 $OUT/tests.log | cut -c1-300
echo "== variants"; FVK_PROBE_LIB=1 timeout 600 python -m pytest scripts/probes/variant_tests.py -q > $OUT/variant.log 2>&1; echo rc=$?; tail -12 $OUT/variant.log | cut -c1-300
echo "== conv3w probe"; FVK_PROBE_LIB=1 timeout 300 python scripts/conv3w_probe.py > $OUT/conv3w_probe.log 2>&1; cat $OUT/conv3w_probe.log | cut -c1-700
echo "== gemm sp shapes"; timeout 300 python scripts/gemm_sp_shapes.py > $OUT/gemm_sp_shapes.log 2>&1; cat $OUT/gemm_sp_shapes.log | cut -c1-300
echo "== sp rank emulation"; timeout 300 python scripts/sp_rank_emulation.py > $OUT/sp_rank_emulation.json 2> $OUT/sp_rank_emulation.err; python - <<'P'
import json
j=json.load(open('gpurun_out/r4b/sp_rank_emulation.json'))
for k,v in j.items(): print(k, v.get('layout'), 'layer_us', v.get('layer_us'), 'speedup', v.get('compute_only_speedup'), {a:round(b,1) for a,b in v.items() if isinstance(b,float) and a not in ('layer_us','compute_only_speedup')})
P
