set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r2final; mkdir -p $OUT
( time timeout 900 python -m pytest tests -q -m gpu -rs ) > "$OUT/pytest_all.log" 2>&1; echo "pytest rc=$? $(tail -4 "$OUT/pytest_all.log" | head -1)"
timeout 200 python -c "import __graft_entry__ as G; G.build(); G.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 500 python bench.py > "$OUT/bench_contract.log" 2> "$OUT/bench_contract.err"; echo "contract rc=$? $(tail -1 "$OUT/bench_contract.log" | cut -c1-1500)"
