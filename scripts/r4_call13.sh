#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
HOOK=1 POISON=512 timeout 300 python scripts/sp_forward_determinism.py 2 6 0 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | cut -c1-330
