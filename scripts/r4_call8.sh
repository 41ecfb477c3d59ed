#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4h; mkdir -p $OUT
timeout 120 python scripts/sp_pack2_debug.py 2>&1 | grep -v amdgpu.ids | cut -c1-300
