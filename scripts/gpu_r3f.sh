#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r3f
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "attn_dense" 2>&1 | tail -8
timeout 600 python scripts/sp_rank_emulation.py > "$OUT/sp_rank_emulation.json" 2> "$OUT/sp.err"; echo "rc=$?"; python - <<'PY'
import json
j=json.load(open("gpurun_out/r3f/sp_rank_emulation.json"))
for k,v in j.items(): print(k, v["layout"], "wg", v["attn_workgroups"], "splits", v["attn_key_splits"], "attn", v["per_op_us"]["self-attention"], "unsplit", v["self_attention_unsplit_us"], "layer", v["layer_us"], "speedup", v.get("compute_only_speedup"))
PY
