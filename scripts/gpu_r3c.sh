#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3c
mkdir -p "$OUT"
timeout 600 python scripts/attn_w64_check.py > "$OUT/w64_check.log" 2>&1; echo "rc=$?"; tail -22 "$OUT/w64_check.log"
