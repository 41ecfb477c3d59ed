#!/bin/bash
# round 4, box visit 3: persistent conv3w (parity, A/B vs one-workgroup-per-tile and vs the 8-wave kernel), pipelined SP diagnostics
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4c; mkdir -p $OUT
echo "== vae tests"; timeout 400 python -m pytest tests/test_gpu_vae.py tests/test_gpu_vae_real.py tests/test_gpu_vae_tiled.py tests/test_gpu_loader.py -q -x > $OUT/vae_tests.log 2>&1; echo rc=$?; tail -15 $OUT/vae_tests.log | cut -c1-400
echo "== conv3w vs 8-wave"; FVK_PROBE_LIB=1 timeout 300 python -m pytest scripts/probes/variant_tests.py -q -k conv3w > $OUT/variant.log 2>&1; echo rc=$?; tail -12 $OUT/variant.log | cut -c1-400
for impl in 0 4 3; do echo "== breakdown impl $impl"; FVK_PROBE_LIB=1 timeout 200 python scripts/vae_conv_breakdown.py --impl $impl > $OUT/vae_breakdown_$impl.log 2>&1; head -8 $OUT/vae_breakdown_$impl.log | cut -c1-200; done
for impl in 0 4; do echo "== probe impl $impl"; VAE_CONV_IMPL=$impl FVK_PROBE_LIB=1 timeout 300 python scripts/conv3w_probe.py > $OUT/conv3w_probe_$impl.log 2>&1; cat $OUT/conv3w_probe_$impl.log | cut -c1-420; done
echo "== sp pipelined"; timeout 600 python -m pytest tests/test_gpu_sp.py -q -k pipelined > $OUT/sp.log 2>&1; echo rc=$?; tail -30 $OUT/sp.log | cut -c1-400
echo "== bench 2 ranks overlap"; FVK_BENCH_SHARED_GPU=1 FVK_SP_OVERLAP=1 timeout 600 python -W always -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --steps 2 --warmup 1 --layers 2 > $OUT/bench2.log 2> $OUT/bench2.err; echo rc=$?; cut -c1-700 $OUT/bench2.log | tail -2; grep -i "disagree\|warn\|error" $OUT/bench2.err | head -10 | cut -c1-600
