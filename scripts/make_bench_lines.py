"""profiles/<round>_bench_lines.md from one sweep directory (scripts/history/round2_sweep.sh): a table + the unedited JSON of every line.
usage: python scripts/make_bench_lines.py gpurun_out/r2sweep3 profiles/r02_bench_lines_final.md "title text" """
import json
import os
import sys

src, dst, title = sys.argv[1], sys.argv[2], sys.argv[3]
ROWS = [("bench_contract", "contract line: cfg2 dense bf16 (reference-kind CPU baseline)"), ("cfg2_sta", "cfg2 sliding-tile (3,3,3)"),
        ("cfg2_vsa", "cfg2 VSA sparsity 0.8"), ("cfg2_fp8", "cfg2 fp8 (tensor) linears"), ("cfg2_fp8_channel", "cfg2 fp8 (channel) linears"),
        ("cfg1", "cfg1 latent 9x64x64"), ("cfg5_dense", "cfg5 129f x 720p, 1.3B geometry, dense"), ("cfg5_vsa_fp8", "cfg5 VSA 0.8 + fp8"),
        ("cfg4", "cfg4 Wan2.2-A14B geometry (one expert), 81f x 720p"), ("vae_cfg2", "VAE decode 81f x 480p"), ("vae_cfg5", "VAE decode 129f x 720p")]


def last_json(path):
    if not os.path.exists(path):
        return None
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith("{"):
            try:
                return json.loads(line)
            except json.JSONDecodeError:
                return None
    return None


out = [f"# {title}", "", f"Source: `{src}` (one box visit, `scripts/history/round2_sweep.sh`).  Boxes differ by a few %.", ""]
py = os.path.join(src, "pytest_all.log")
if os.path.exists(py):
    tail = [l for l in open(py).read().splitlines() if " passed" in l or " failed" in l]
    if tail:
        out += [f"`-m gpu` suite on this visit: {tail[-1].strip('= ')}", ""]
out += ["| run | ms / step | step TF (algorithmic) | dominant kernel TF (frac of 2.5 PF) | notes |", "|---|---:|---:|---:|---|"]
blobs = []
for key, label in ROWS:
    d = last_json(os.path.join(src, key + ".log"))
    if d is None:
        continue
    r = d.get("roofline", {})
    notes = []
    if r.get("effective_clock_ghz"):
        notes.append(f"effective clock {r['effective_clock_ghz']} GHz -> {r.get('frac_of_clock_limited_peak')} of the clock-limited peak")
    if r.get("traffic"):
        notes.append(f"fabric traffic {r['traffic'] / 1e9:.2f} GB / launch ({r.get('traffic_source')})")
    cb = d.get("cpu_baseline")
    if isinstance(cb, dict) and cb.get("value") is not None:
        notes.append(f"CPU baseline ({cb.get('kind')}, {cb.get('cores')} threads): {cb['value']} {cb.get('unit')}")
    out.append(f"| {label} | **{d['ms_per_step']:.1f}** | {d.get('step_tflops')} | {r.get('achieved')} ({r.get('frac')}) | {'; '.join(notes)} |")
    blobs.append((key, d))
c = last_json(os.path.join(src, "causal_480p.log"))
if c:
    out += ["", "Causal Wan2.1-1.3B rollout (scripts/causal_bench.py): `" + json.dumps(c) + "`"]
out += ["", "## unedited JSON lines", ""]
for key, d in blobs:
    out += [f"### {key}", "", "```", json.dumps(d), "```", ""]
open(dst, "w").write("\n".join(out) + "\n")
print(f"wrote {dst}: {len(blobs)} lines")
