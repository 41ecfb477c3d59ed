#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4u; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ref_triton.py tests/test_gpu_model.py -q -k "vsa or union or block_sparse or sparse" > $OUT/tests.log 2>&1; echo rc=$?; tail -8 $OUT/tests.log | cut -c1-300
timeout 300 python scripts/vsa_union_ab.py > $OUT/vsa_union_ab.log 2>&1; cat $OUT/vsa_union_ab.log | grep -v amdgpu.ids | cut -c1-300
timeout 300 python bench.py --attention vsa --steps 5 --warmup 2 --no-cpu-baseline --no-vae > $OUT/bench_vsa.log 2>$OUT/bench_vsa.err; tail -1 $OUT/bench_vsa.log | cut -c1-900
