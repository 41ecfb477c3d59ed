#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python scripts/sta_w64_ab.py 2>&1 | tail -2
timeout 300 python scripts/sta_w64_ab.py 18 48 80 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_boundary.py tests/test_gpu_ref_triton.py tests/test_gpu_fullsize.py tests/test_gpu_model.py -m gpu -q -k "sta or tile_lists or sliding or attn" 2>&1 | tail -8
