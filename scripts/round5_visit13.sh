#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r5v13
mkdir -p "$OUT"
for args in "6 fp8_channel cfg2" "8 bf16 cfg1" "6 fp8 cfg2"; do
  timeout 300 python scripts/step_flags_ab.py $args > "$OUT/flags_$(echo $args | tr ' ' '_').log" 2>&1; echo "flags [$args] rc=$?"
  tail -1 "$OUT/flags_$(echo $args | tr ' ' '_').log" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['quant'], j['config'], j['median_ms'], j['bit_identical_to_shipped'])"
done
