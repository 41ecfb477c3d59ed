#!/bin/bash
# round 3: attn_w16 (16x16x32 MFMAs) correctness + A/B against attn_w64 / attn_pp2
mkdir -p gpurun_out/r3m; cd /root/repo
CHECK_IMPL=300 timeout 900 python scripts/attn_w64_check.py 200 300 200 300 > gpurun_out/r3m/check.log 2>&1
cat gpurun_out/r3m/check.log | tail -20
