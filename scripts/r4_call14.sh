#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do OPS=1 HOOK=1 timeout 300 python scripts/sp_forward_determinism.py 2 6 0 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | grep -v " 0 of .* 0 of " | cut -c1-260 | head -14; done
