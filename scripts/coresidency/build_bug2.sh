#!/bin/bash
# scripts/probes/libfvk_bug2.so (co-residency study): libfvk_bug.so with gemm_w1.hip compiled -DFVK_W1_BUILTIN_MFMA=1 (its MFMAs as
# compiler builtins instead of inline asm).  Needs libfvk_bug.so's objects (python -c "from fastvideo_amd import _build; _build.build_bug()").
set -e
cd "$(dirname "$0")/../.."
B=scripts/probes/build_bug
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -mllvm -pragma-unroll-threshold=100000 -DFVK_PROBE_BUILD=1 \
      -DFVK_NO_REGISTER_CLAIM=1 -DFVK_W1_BUILTIN_MFMA=1 -I fastvideo_amd/csrc -c fastvideo_amd/csrc/gemm_w1.hip -o $B/gemm_w1_builtin.o
OBJS=$(ls $B/*.o | grep -v "/gemm_w1.o$" | grep -v "/gemm_w1_builtin.o$")
hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/probes/libfvk_bug2.so $OBJS $B/gemm_w1_builtin.o
ls -la scripts/probes/libfvk_bug2.so
