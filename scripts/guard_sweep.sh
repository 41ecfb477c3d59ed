#!/bin/bash
# Every GPU test file under the guard-page device allocator (scripts/probes/guard_alloc.cpp, tests/_guard.py): one process per file, an unmapped
# page behind every tensor, so an out-of-bounds access of any kernel is a GPU fault.  Report of round 6: profiles/r06d_guard_page_runs.md.
# FVK_TRACE_CALLS=1 in front of a failing file names every C-ABI call and runs it to completion (the last name printed is the culprit).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/guard_sweep; mkdir -p $OUT
timeout 300 python scripts/guard_selftest.py > $OUT/selftest.log 2>&1; echo "selftest rc=$? $(tail -1 $OUT/selftest.log)"
timeout 120 python scripts/guard_selftest.py oob > $OUT/selftest_oob.log 2>&1; echo "deliberate overrun rc=$? $(grep -i 'fault\|NO FAULT' $OUT/selftest_oob.log | head -1)"
for T in tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_sp.py tests/test_gpu_fullgeom.py tests/test_gpu_fp8.py tests/test_gpu_sched.py \
         tests/test_gpu_causal.py tests/test_gpu_loader.py tests/test_gpu_reference_model.py tests/test_gpu_vae.py tests/test_gpu_vae_tiled.py \
         tests/test_gpu_vae_real.py tests/test_gpu_boundary.py tests/test_gpu_ref_triton.py tests/test_gpu_fullsize.py tests/test_gpu_rccl_single_rank.py; do
  [ -f $T ] || continue
  N=$(basename $T .py)
  FVK_GUARD_ALLOC=1 timeout 1500 python -m pytest $T -q > $OUT/$N.log 2>&1; echo "guard $T rc=$? $(tail -1 $OUT/$N.log | cut -c1-150)"
  grep -i "memory access fault" $OUT/$N.log | sort | uniq -c | head -3
done
FVK_GUARD_ALLOC=1 FVK_GUARD_MODE=front timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ref_triton.py -q > $OUT/front.log 2>&1; echo "front placement rc=$? $(tail -1 $OUT/front.log | cut -c1-150)"
