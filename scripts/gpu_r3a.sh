#!/bin/bash
# round 3, GPU visit A: the new / changed parity tests, the attention timing ablations (measurement build), the contract bench line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3a
mkdir -p "$OUT"
echo "== new tests"
timeout 1500 python -m pytest tests/test_gpu_vae_real.py tests/test_gpu_fullgeom.py tests/test_gpu_reference_model.py tests/test_gpu_sched.py \
    tests/test_gpu_probe_variants.py tests/test_gpu_sp.py tests/test_gpu_boundary.py tests/test_gpu_ref_triton.py tests/test_gpu_kernels.py \
    -m gpu -q -s --durations=15 > "$OUT/pytest_new.log" 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error" "$OUT/pytest_new.log" | tail -5; grep -E "^(FAILED|ERROR)" "$OUT/pytest_new.log" | head -20
echo "== ablations"
AB_ROUNDS=5 timeout 300 python scripts/attn_ab.py 0 121 122 123 124 125 127 > "$OUT/attn_ablate.log" 2>&1; echo "rc=$?"; cat "$OUT/attn_ablate.log" | tail -12
echo "== bench"
timeout 600 python bench.py > "$OUT/bench.log" 2> "$OUT/bench.err"; echo "bench rc=$?"; tail -1 "$OUT/bench.log" | cut -c1-3000
