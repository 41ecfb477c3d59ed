#!/bin/bash
# round 5, first visit: the new parity tests (cfg2-geometry sta / vsa / fp8, STA text path, torch.library ops, V^T GEMM), the V^T-GEMM A/B,
# a de-mixed rocprof summary of the contract bench (no cfg_step / vae legs), the VSA block-sparse counter passes, the contract line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r5v1
mkdir -p "$OUT"
python -c "
import ctypes, os
ctypes.CDLL(os.path.abspath('fastvideo_amd/libfvk_amd.so')); print('lib loads')
" || exit 1
( time timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm_vt or v_transpose or gemm_bias or gemm_epilogues" -s ) > "$OUT/pytest_vt.log" 2>&1; echo "pytest vt rc=$?"; tail -4 "$OUT/pytest_vt.log" | cut -c1-400
( time timeout 2400 python -m pytest tests/test_gpu_fullgeom.py tests/test_gpu_boundary.py tests/test_gpu_sched.py "tests/test_gpu_ref_triton.py" -m gpu -q -rs -s ${VAE_FULL:+tests/test_gpu_vae_real.py} ) > "$OUT/pytest_new.log" 2>&1; echo "pytest new rc=$?"; tail -8 "$OUT/pytest_new.log" | cut -c1-400
grep -E "cfg2 |vsa block selection|STA .* text|81 frames|per-frame" "$OUT/pytest_new.log" | cut -c1-300
timeout 600 python scripts/vt_gemm_ab.py > "$OUT/vt_gemm_ab.log" 2>&1; echo "vt ab rc=$?"; tail -2 "$OUT/vt_gemm_ab.log" | cut -c1-1500
bash scripts/prof.sh r5v1 --no-cfg-step --no-vae --no-power-trace 2>&1 | grep -v "distribution\|at::native" | tail -22 | cut -c1-220
bash scripts/vsa_pmc_r5.sh > "$OUT/vsa_pmc.log" 2>&1; echo "vsa pmc rc=$?"; tail -45 "$OUT/vsa_pmc.log" | cut -c1-200
timeout 900 python bench.py > "$OUT/bench.log" 2> "$OUT/bench.err"; echo "bench rc=$?"; tail -1 "$OUT/bench.log" | cut -c1-3500
