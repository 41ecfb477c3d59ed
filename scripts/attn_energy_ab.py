"""Round 5: an ENERGY account of the dense attention kernel at the power cap.

At the cap the socket power is pinned, so the wall time of a power-bound launch IS its energy: (P_cap - P_idle) * t = sum over instruction
classes of (events x energy per event).  This script measures, each for ~2.5 s of back-to-back launches with the amdsmi sampler running
(scripts/power_trace.py: socket power, shader clock):
  * the registers-only MFMA loops of scripts/probes/mfma_energy.cpp — both bf16 shapes, operand data zeros / near-constant / uniform random /
    normal-like: what the cap leaves of the matrix pipe ALONE as a function of operand toggling;
  * attn_w16 (the shipped dense kernel) at the contract shape on randn and on zero operands, its timing ablations (measurement build:
    no LDS-DMA, no softmax VALU, neither + no barrier, no barrier alone) and attn_w64 / attn_pp2 for reference.
Output: one JSON line per measurement + a summary with the time (= energy) shares.  usage: python scripts/attn_energy_ab.py"""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")
import importlib.util, json, os, subprocess, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import torch
from fastvideo_amd import ops
spec = importlib.util.spec_from_file_location("power_trace", os.path.join(HERE, "power_trace.py"))
pt = importlib.util.module_from_spec(spec); spec.loader.exec_module(pt)
SECONDS = float(os.environ.get("ENERGY_SECONDS", "2.5"))


def sampled(fn_loop):
    """fn_loop() runs for SECONDS and returns (launches, gpu_ms_total); sampled from 0.5 s in (the clock has settled by then)."""
    ps = pt.PowerSampler(20.0).start()
    t0 = time.time()
    n, ms = fn_loop()
    t1 = time.time()
    sm = pt.summarize(ps.stop(), t0 + 0.5, t1, "amdsmi")
    g = lambda k: round(sm[k]["mean"], 1) if sm.get(k) else None
    return n, ms, {"power_w": g("power_w"), "sclk_mhz": g("sclk_mhz"), "sclk_min_over_xcds_mhz": g("sclk_min_over_xcds_mhz")}


results = []
exe = os.path.join(HERE, "probes", "mfma_energy")
if os.path.exists(exe):
    for shape in (1, 0):
        for data, dname in ((0, "zeros"), (1, "near-constant (round 3's probe)"), (2, "uniform random"), (3, "normal-like")):
            def loop():
                r = subprocess.run([exe, str(shape), str(data), str(SECONDS)], capture_output=True, text=True, timeout=60)
                j = json.loads(r.stdout.strip().splitlines()[-1])
                return j["launches"], j["ms_per_launch"] * j["launches"], j
            ps = pt.PowerSampler(20.0).start()
            t0 = time.time()
            n, ms, j = loop()
            t1 = time.time()
            sm = pt.summarize(ps.stop(), t0 + 0.8, t1 - 0.1, "amdsmi")
            g = lambda k: round(sm[k]["mean"], 1) if sm.get(k) else None
            rec = {"what": f"registers-only MFMA loop {j['shape']}, operands {dname}", "tflops": j["tflops"], "power_w": g("power_w"), "sclk_mhz": g("sclk_mhz"),
                   "matrix_pipe_peak_at_that_clock_tf": round(2500 * (g("sclk_mhz") or 2400) / 2400, 1)}
            results.append(rec); print(json.dumps(rec), flush=True)
else:
    print(json.dumps({"warning": "scripts/probes/mfma_energy not built"}), flush=True)

B, S, H, D = 1, 32760, 12, 128
gen = torch.Generator(device="cuda").manual_seed(0)
q, k, v = (torch.randn((B, S, H, D), generator=gen, device="cuda").bfloat16() for _ in range(3))
z = torch.zeros_like(q)
FLOP = 4.0 * S * S * H * D
variants = [(300, "attn_w16 shipped", (q, k, v)), (300, "attn_w16 shipped, ZERO operands", (z, z, z)), (311, "attn_w16 without the LDS-DMA pieces in the loop", (q, k, v)),
            (314, "attn_w16 without the softmax VALU in the loop", (q, k, v)), (312, "attn_w16 without the stage barrier", (q, k, v)),
            (317, "attn_w16 without DMA, softmax VALU and barrier (MFMA + LDS fragment reads + loop control)", (q, k, v)),
            (200, "attn_w64 (32x32x16 MFMAs)", (q, k, v)), (99, "attn_pp2 (8 waves, 32x32x16)", (q, k, v)), (300, "attn_w16 shipped (again)", (q, k, v))]
for impl, name, (q_, k_, v_) in variants:
    ops.set_tunable("attn_impl", impl)
    vt = ops.v_transpose(v_)
    for _ in range(5): ops.attn_dense(q_, k_, vt=vt, layout="bshd")
    torch.cuda.synchronize()

    def loop():
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0, n = time.time(), 0
        s.record()
        while time.time() - t0 < SECONDS:
            for _ in range(20): ops.attn_dense(q_, k_, vt=vt, layout="bshd")
            n += 20
            torch.cuda.synchronize()
        e.record(); torch.cuda.synchronize()
        return n, s.elapsed_time(e)
    n, ms, pw = sampled(loop)
    rec = {"what": name, "attn_impl": impl, "ms_per_launch": round(ms / n, 4), "tflops": round(FLOP / (ms / n) / 1e9, 1), **pw}
    results.append(rec); print(json.dumps(rec), flush=True)
ops.set_tunable("attn_impl", 0)
# idle power
time.sleep(1.0)
ps = pt.PowerSampler(20.0).start(); t0 = time.time(); time.sleep(2.0); sm = pt.summarize(ps.stop(), t0 + 0.5, time.time(), "amdsmi")
idle = round(sm["power_w"]["mean"], 1) if sm.get("power_w") else None
byname = {r["what"]: r for r in results}
t = lambda n_: byname[n_]["ms_per_launch"]
ship = t("attn_w16 shipped")
summary = {"idle_power_w": idle, "shipped_ms": ship,
           "time_share_lds_dma": round((ship - t("attn_w16 without the LDS-DMA pieces in the loop")) / ship, 4),
           "time_share_softmax_valu": round((ship - t("attn_w16 without the softmax VALU in the loop")) / ship, 4),
           "time_share_barrier": round((ship - t("attn_w16 without the stage barrier")) / ship, 4),
           "time_share_mfma_lds_reads_loop": round(t("attn_w16 without DMA, softmax VALU and barrier (MFMA + LDS fragment reads + loop control)") / ship, 4),
           "zero_operand_speedup": round(ship / t("attn_w16 shipped, ZERO operands"), 4)}
print(json.dumps({"summary": summary}), flush=True)
