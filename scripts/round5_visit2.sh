#!/bin/bash
# round 5, second visit: the re-stated fp8 / STA full-geometry tests, the model tests with the fused cross-attention residual, the full-length VAE
# parity (when its fixture exists), the energy account of the dense attention kernel, the co-residency bug studies, the step A/B of the two
# round-5 host-side changes, the contract line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r5v2
mkdir -p "$OUT"
python -c "
import ctypes, os
for p in ('fastvideo_amd/libfvk_amd.so', 'scripts/probes/libfvk_probe.so', 'scripts/probes/libfvk_bug.so'):
    ctypes.CDLL(os.path.abspath(p)); print('loads', p)
" || exit 1
VAE_T=""; [ -f tests/golden/vae_full_480p.pt ] && VAE_T="tests/test_gpu_vae_real.py"
( time timeout 2400 python -m pytest tests/test_gpu_fullgeom.py tests/test_gpu_model.py tests/test_gpu_kernels.py $VAE_T -m gpu -q -rs -s -k "not contract_vs_reference" ) > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -6 "$OUT/pytest.log" | cut -c1-400
grep -E "cfg2 (fp8|sta)|81 frames|per-frame|ratio of means" "$OUT/pytest.log" | cut -c1-420
timeout 600 python scripts/attn_energy_ab.py > "$OUT/attn_energy_ab.log" 2>&1; echo "energy rc=$?"; grep -v "^W\|amdgpu.ids" "$OUT/attn_energy_ab.log" | cut -c1-330
timeout 300 scripts/probes/coresidency_probe 40 > "$OUT/coresidency_synthetic.log" 2>&1; echo "synthetic rc=$?"; python - "$OUT/coresidency_synthetic.log" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    try: j = json.loads(ln)
    except Exception: print(ln.strip()[:200]); continue
    print(f"{j['wrong']:>10} wrong (lo {j['wrong_low_half']}, hi {j['wrong_high_half']}) | agg {j['aggressor_ms']:6.1f} ms vic {j['victim_ms']:6.1f} ms | {j['aggressor'][:70]:70s} | {j['victim'][:50]}")
PY
FVK_PROBE_LIB=bug timeout 900 python scripts/coresidency_matrix.py 120 > "$OUT/coresidency_matrix.log" 2>&1; echo "matrix rc=$?"; grep -v '"wrong": 0,' "$OUT/coresidency_matrix.log" | grep -v "^W\|amdgpu.ids" | cut -c1-600 | tail -40
timeout 600 python scripts/vt_gemm_ab.py > "$OUT/vt_gemm_ab.log" 2>&1; echo "vt ab rc=$?"; head -1 "$OUT/vt_gemm_ab.log" | cut -c1-700
timeout 900 python scripts/step_flags_ab.py 6 > "$OUT/step_flags_ab.log" 2>&1; echo "flags ab rc=$?"; tail -1 "$OUT/step_flags_ab.log" | cut -c1-1500
timeout 900 python bench.py > "$OUT/bench.log" 2> "$OUT/bench.err"; echo "bench rc=$?"; tail -1 "$OUT/bench.log" | cut -c1-1800
