set -x
timeout 900 python -m pytest tests/test_gpu_vae.py tests/test_gpu_vae_tiled.py -x -q 2>&1 | tail -5
for F in 1 2 4 7 20; do
  timeout 300 python bench.py --stage vae --steps 3 --warmup 1 --no-cpu-baseline --vae-frames-per-pass $F 2>/dev/null | tail -1 > gpurun_out/vae_f$F.json
  python -c "import json;d=json.load(open('gpurun_out/vae_f$F.json'));print($F,d['ms_per_step'],d['step_tflops'],d['roofline']['achieved'],d['peak_mem_gb'])"
done
