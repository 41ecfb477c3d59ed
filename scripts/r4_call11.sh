#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2 3; do BRIEF=1 POISON=512 timeout 300 python scripts/sp_forward_determinism.py 3 6 0 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | cut -c1-250; done
for i in 1 2; do BRIEF=1 POISON=512 timeout 300 python scripts/sp_forward_determinism.py 2 6 0 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | cut -c1-250; done
