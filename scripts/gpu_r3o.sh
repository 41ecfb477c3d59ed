#!/bin/bash
# round 3: where attn_w16 overtakes attn_pp2 on short key axes
mkdir -p gpurun_out/r3o; cd /root/repo
for L in 512 1024 1536 2048 3072 4096; do echo -n "L=$L "; CROSS_L=$L timeout 200 python scripts/cross_attn_ab.py 2>/dev/null | tail -1; done | tee gpurun_out/r3o/short_keys.log
