export TMPDIR=/tmp
O=gpurun_out/r6v18; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fp8.py -m gpu -x -q > $O/fp8_tests.log 2>&1; echo "fp8 tests rc=$?"; tail -3 $O/fp8_tests.log
timeout 900 python -m pytest tests/test_gpu_fullgeom.py tests/test_gpu_model.py tests/test_gpu_loader.py -m gpu -x -q -k "fp8" > $O/fp8_model_tests.log 2>&1; echo "fp8 model tests rc=$?"; tail -3 $O/fp8_model_tests.log
for q in fp8_channel none fp8_channel none; do
  if [ $q = none ]; then A=""; else A="--quant $q"; fi
  timeout 900 python bench.py $A --steps 10 --warmup 3 --no-cpu-baseline --no-vae --no-cfg-step --no-matrix-ceiling > $O/bench_$q.json 2> $O/bench_$q.err; python -c "
import json;d=json.loads(open('$O/bench_$q.json').read().strip().splitlines()[-1]);print('$q', d['ms_per_step'])"
done
