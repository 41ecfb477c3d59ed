set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/vae3; mkdir -p $OUT
timeout 400 python bench.py --stage vae --steps 5 --warmup 2 > "$OUT/vae_cfg2.log" 2> "$OUT/vae_cfg2.err"; echo "vae cfg2 rc=$? $(tail -1 "$OUT/vae_cfg2.log" | cut -c1-900)"
timeout 400 python bench.py --stage vae --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/vae_cfg5.log" 2>&1; echo "vae cfg5 rc=$? $(tail -1 "$OUT/vae_cfg5.log" | cut -c1-500)"
timeout 200 python scripts/vae_conv_breakdown.py --frames 33 --h 90 --w 160 2>&1 | tail -15
timeout 200 python scripts/vae_bench.py --mode stream 2>&1 | tail -2
timeout 200 python scripts/vae_bench.py --mode tiled 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_vae" -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --stage vae > "$OUT/prof_vae.log" 2>&1 < /dev/null; echo "prof rc=$?"
F=$(find "$OUT/prof_vae" -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && python scripts/condense_prof.py "$F" "$OUT/vae_kernel_stats.csv"; find "$OUT/prof_vae" -name "*kernel_trace.csv" -delete
