#!/bin/bash
# scripts/probes/libfvk_bug_s<N>.so (co-residency study): libfvk_bug.so with gemm_w1.hip compiled -DW1_STRIP=<N> (the aggressor with parts
# of its main loop removed: 1 no LDS-DMA, 2 no fragment reads, 4 no barriers, 8 no MFMAs; sums combine).  Needs libfvk_bug.so's objects.
set -e
cd "$(dirname "$0")/../.."
B=scripts/probes/build_bug
for N in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -mllvm -pragma-unroll-threshold=100000 -DFVK_PROBE_BUILD=1 \
        -DFVK_NO_REGISTER_CLAIM=1 -DW1_STRIP=$N -I fastvideo_amd/csrc -c fastvideo_amd/csrc/gemm_w1.hip -o $B/gemm_w1_s$N.o &
done
wait
for N in "$@"; do
  OBJS=$(ls $B/*.o | grep -v "/gemm_w1.o$" | grep -v "/gemm_w1_builtin.o$" | grep -v "/gemm_w1_s[0-9]*.o$")
  hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/probes/libfvk_bug_s$N.so $OBJS $B/gemm_w1_s$N.o
  ls -la scripts/probes/libfvk_bug_s$N.so
done
