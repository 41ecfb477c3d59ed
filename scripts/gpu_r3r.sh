#!/bin/bash
mkdir -p gpurun_out/r3r; cd /root/repo
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "kernel_choice or attn_dense" 2>&1 | tail -5
timeout 600 python bench.py --no-vae --no-cpu-baseline > gpurun_out/r3r/bench.json 2> gpurun_out/r3r/bench.err; tail -1 gpurun_out/r3r/bench.json | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['achieved'], j['roofline'].get('kernel_choice'), j['roofline']['kernel'][:20])"
tail -3 gpurun_out/r3r/bench.err
