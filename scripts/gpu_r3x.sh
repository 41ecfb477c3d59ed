#!/bin/bash
# the in-place attention kernel choice under sequence parallelism: 2 ranks sharing the one GPU (host-staged gloo exchange: numbers invalid, flow exercised)
mkdir -p gpurun_out/r3x; cd /root/repo
timeout 600 python -m pytest tests/test_gpu_sp.py -m gpu -q 2>&1 | tail -3
FVK_BENCH_SHARED_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-vae 2> gpurun_out/r3x/sp2.err | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['config']['parallelism'], j['roofline'].get('kernel_choice'))"
tail -3 gpurun_out/r3x/sp2.err
