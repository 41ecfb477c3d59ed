#!/bin/bash
# round 4, box visit 4: persistent conv3w A/B (vs one workgroup per tile, vs the 8-wave kernel) + tile probe, pipelined SP (single stream) checks
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r4d; mkdir -p $OUT
python -c "import ctypes,torch; ctypes.CDLL('fastvideo_amd/libfvk_amd.so'); ctypes.CDLL('scripts/probes/libfvk_probe.so'); print('both libraries load')" || exit 1
echo "== conv3w vs 8-wave"; FVK_PROBE_LIB=1 timeout 300 python -m pytest scripts/probes/variant_tests.py -q -k "conv3w or w1n" > $OUT/variant.log 2>&1; echo rc=$?; tail -6 $OUT/variant.log | cut -c1-400
for impl in 0 4 3; do echo "== breakdown impl $impl"; FVK_PROBE_LIB=1 timeout 200 python scripts/vae_conv_breakdown.py --impl $impl > $OUT/vae_breakdown_$impl.log 2>&1; head -8 $OUT/vae_breakdown_$impl.log | cut -c1-200; done
for impl in 0 4; do echo "== probe impl $impl"; VAE_CONV_IMPL=$impl FVK_PROBE_LIB=1 timeout 300 python scripts/conv3w_probe.py > $OUT/conv3w_probe_$impl.log 2>&1; cat $OUT/conv3w_probe_$impl.log | cut -c1-420; done
echo "== sp pipelined + masks"; timeout 600 python -m pytest tests/test_gpu_sp.py tests/test_gpu_boundary.py -q -k "pipelined_exchange_equals_sp1 and 12 or key_padding" > $OUT/sp.log 2>&1; echo rc=$?; tail -30 $OUT/sp.log | cut -c1-400
echo "== bench 2 ranks overlap"; FVK_BENCH_SHARED_GPU=1 FVK_SP_OVERLAP=1 timeout 600 python -W always -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --steps 2 --warmup 1 --layers 2 > $OUT/bench2.log 2> $OUT/bench2.err; echo rc=$?; python - <<'P'
import json
for ln in open('gpurun_out/r4d/bench2.log'):
    if ln.startswith('{'):
        j=json.loads(ln); print(j['config']['parallelism'], j['ms_per_step'], j.get('exchange',{}).get('per_kind'))
P
grep -i "disagree\|error" $OUT/bench2.err | head -5 | cut -c1-600
