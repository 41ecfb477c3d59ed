#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for cfg in "3 6 1" "3 6 0" "2 12 0"; do timeout 500 python scripts/sp_forward_determinism.py $cfg 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | cut -c1-250; done
