#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r5v14
mkdir -p "$OUT"
( time timeout 500 python -m pytest tests/test_gpu_vae_real.py tests/test_gpu_sp.py -m gpu -q -s -k "full_length or sparse" ) > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; grep -E "frames of|per-frame|passed|failed" "$OUT/pytest.log" | cut -c1-400
