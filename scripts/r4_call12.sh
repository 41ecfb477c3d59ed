#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
POISON=512 timeout 300 python scripts/sp_forward_determinism.py 1 6 0 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | cut -c1-330
POISON=512 timeout 300 python scripts/sp_forward_determinism.py 1 12 0 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | cut -c1-330
