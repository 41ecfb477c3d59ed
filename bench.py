#!/usr/bin/env python
"""bench.py — the contract benchmark: one "step" = one Wan2.1-T2V-1.3B DiT forward (the per-step denoising hot path,
SURVEY.md §8a2) over a synthetic 81f x 832 x 480 latent ([1,16,21,60,104] -> 32 760 tokens), bf16, random-init weights
of the 1.3B architecture, dense attention (BASELINE.json configs[1]).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

N > 1 shards the token axis over the N GPUs with 2-D Ulysses sequence parallelism (fastvideo_amd/distributed.py; RCCL
all-to-all over xGMI), i.e. the SAME forward split N ways -> "strong" scaling.  Rank 0 prints ONE JSON line.
`value` = latent tokens per second of the whole job; inputs are resident in HBM before the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (guides/MI355X_MICROARCH.md); never the 2:1-sparse figure


def cpu_baseline_reference(cfg, latent_shape, S, L_text, sample_layers=2):
    """The REFERENCE ITSELF: hao-ai-lab/FastVideo's own ``WanTransformer3DModel`` through its
    ``SDPABackend`` (fastvideo/models/dits/wanvideo.py, fastvideo/attention/backends/sdpa.py), imported with the App. A harness from
    /root/reference or from the copy staged by oracle/stage_ref.py, fp32 on the host cores.  BOUNDED sample of the same workload:
    the full cfg latent, ``sample_layers`` of the model's transformer blocks; each block is timed by forward hooks, and the forward
    is extrapolated as (mean block time) x num_layers + (measured time outside the blocks).  Returns None when no reference tree
    is present (the caller then falls back to the port)."""
    from oracle import ref_loader as R
    if not R.available():
        return None
    cores = os.cpu_count() or 1
    threads = min(cores, 64)  # torch CPU GEMM/SDPA stop scaling (and oversubscribe) far below 256 threads
    torch.set_num_threads(threads)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    model = R.build_wan(num_heads=cfg.num_heads, head_dim=cfg.head_dim, ffn_dim=cfg.ffn_dim, num_layers=sample_layers,
                        text_dim=cfg.text_dim, seed=0)
    from fastvideo.forward_context import set_forward_context
    g = torch.Generator().manual_seed(1)
    x = torch.randn(latent_shape, generator=g)
    ctx = torch.randn((1, L_text, cfg.text_dim), generator=g)
    marks = []
    for blk in model.blocks:
        blk.register_forward_pre_hook(lambda m, a: marks.append(("in", time.perf_counter())))
        blk.register_forward_hook(lambda m, a, o: marks.append(("out", time.perf_counter())))

    def run(lat):
        marks.clear()
        t0 = time.perf_counter()
        with torch.no_grad(), set_forward_context(current_timestep=0, attn_metadata=None):
            model(hidden_states=lat, encoder_hidden_states=ctx, timestep=torch.tensor([500]))
        return time.perf_counter() - t0

    run(torch.randn((1, latent_shape[1], 2, 8, 8), generator=g))  # warm-up: thread pool, oneDNN primitives
    total = run(x)
    ins = [t for k, t in marks if k == "in"]
    outs = [t for k, t in marks if k == "out"]
    blocks = [b - a for a, b in zip(ins, outs)]
    per_block = sum(blocks) / len(blocks)
    per_forward = per_block * cfg.num_layers + (total - sum(blocks))
    return dict(value=round(S / per_forward, 2), unit="latent-tokens/s", cores=threads, kind="reference",
                sample=f"the reference's WanTransformer3DModel (fastvideo/models/dits/wanvideo.py, SDPABackend, torch CPU fp32, {threads} threads of "
                       f"{cores} cores) with {sample_layers} of {cfg.num_layers} blocks, one forward on the full latent {list(latent_shape)}: "
                       f"{total:.2f} s, blocks {', '.join(f'{b:.2f}' for b in blocks)} s; forward = mean block x {cfg.num_layers} + "
                       f"{total - sum(blocks):.2f} s outside the blocks", ms_per_step=round(per_forward * 1e3, 1),
                reference_root=("live checkout" if R.REF_ROOT.startswith("/root/reference") else "oracle/_ref (staged by oracle/stage_ref.py)"),
                staged_tree_sha256=_staged_tree_sha())


def _staged_tree_sha():
    """Digest of the staged reference tree (oracle/_ref/MANIFEST.json: sha256 over every path:sha256 line) — makes a
    "reference"-kind baseline reproducible from the commit + the reference revision, although oracle/_ref itself is untracked."""
    try:
        from oracle import stage_ref
        man = stage_ref.manifest()
        return man.get("tree_sha256") if man else None
    except Exception:  # noqa: BLE001
        return None


def cpu_baseline_port(cfg, S, L_text, budget_s=25.0):
    """Reference-algorithm CPU baseline ("port" = oracle/wan_oracle.py, the restatement of the reference eager path that is
    pinned bit-exact against the real reference) on a BOUNDED sample: ONE of the 30 transformer blocks, fp32 weights and
    activations (the reference's CPU harness dtype), at the largest sequence length whose block fits the time
    budget, scaled to the full S by algorithmic FLOPs and x30 layers."""
    from oracle import wan_oracle as W
    from fastvideo_amd.wan_config import WanConfig, algorithmic_flops, random_state_dict
    one = WanConfig(cfg.name, cfg.num_heads, cfg.head_dim, cfg.ffn_dim, 1, cfg.text_dim)
    sd = {k: v.float() for k, v in random_state_dict(one, seed=0, device="cpu").items()}
    orc = W.WanOracle(sd, num_heads=cfg.num_heads)
    cores = os.cpu_count() or 1
    threads = min(cores, 64)  # torch CPU GEMM/SDPA stop scaling (and oversubscribe) far below 256 threads
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(1)
    cos_full, sin_full = W.rope_tables((21, 30, 52), cfg.head_dim)

    def run(Sx):
        x = torch.randn((1, Sx, cfg.dim), generator=g)
        ctx = torch.randn((1, L_text, cfg.dim), generator=g)
        tproj = torch.randn((1, 6, cfg.dim), generator=g) * 0.1
        t0 = time.perf_counter()
        with torch.no_grad():
            orc.block(0, x, ctx, tproj, cos_full[:Sx], sin_full[:Sx])
        return time.perf_counter() - t0

    run(256)  # warm-up: thread pool, oneDNN primitives
    Sx, t = 1024, run(1024)
    while Sx * 2 <= S:
        fl_ratio = algorithmic_flops(one, Sx * 2, L_text)["total"] / algorithmic_flops(one, Sx, L_text)["total"]
        if t * fl_ratio > budget_s:
            break
        Sx *= 2
        t = run(Sx)
    if Sx * 2 > S and t * (algorithmic_flops(one, S, L_text)["total"] / algorithmic_flops(one, Sx, L_text)["total"]) <= budget_s:
        Sx, t = S, run(S)
    scale = algorithmic_flops(one, S, L_text)["total"] / algorithmic_flops(one, Sx, L_text)["total"]
    per_forward = t * scale * cfg.num_layers
    return dict(value=round(S / per_forward, 2), unit="latent-tokens/s", cores=threads, kind="port",
                sample=f"1 of {cfg.num_layers} transformer blocks at S={Sx} in {t:.2f} s (oracle/wan_oracle.py, torch CPU fp32, "
                       f"{threads} threads of {cores} cores), scaled by algorithmic FLOPs to S={S} and x{cfg.num_layers} layers",
                ms_per_step=round(per_forward * 1e3, 1))



def cpu_baseline(cfg, latent_shape, S, L_text, kind="auto"):
    if kind in ("auto", "reference"):
        try:
            out = cpu_baseline_reference(cfg, latent_shape, S, L_text)
            if out is not None:
                return out
            if kind == "reference":
                return {"error": "no reference tree (neither /root/reference nor oracle/_ref/reference)"}
        except Exception as ex:  # noqa: BLE001 - fall back to the port, but say why
            port = cpu_baseline_port(cfg, S, L_text)
            port["reference_error"] = repr(ex)[:300]
            return port
    return cpu_baseline_port(cfg, S, L_text)


def attention_traffic_from_profiles(kernel="attn_w16"):
    """HBM-side bytes per launch of the dominant kernel from the newest committed PMC pass (profiles/*pmc_<kernel>*.json, written by
    scripts/pmc_traffic.sh from `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` passes with the gfx950 x2 FETCH correction; counters cannot be
    collected inside the timed run).  A pass only counts if it recorded the sha256 of the kernel source it measured and that still matches
    the source in the tree: a stale number is reported as null, never silently."""
    import glob
    import hashlib
    src = os.path.join(ROOT, "fastvideo_amd", "csrc", kernel + ".hip")
    cur = hashlib.sha256(open(src, "rb").read()).hexdigest() if os.path.exists(src) else None
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*pmc_{kernel}*.json"))):
        try:
            j = json.load(open(f))
        except Exception:  # noqa: BLE001
            continue
        if j.get("kernel_source_sha256") == cur and j.get("traffic_bytes_per_launch"):
            best = (j["traffic_bytes_per_launch"], os.path.basename(f), j.get("GRBM_GUI_ACTIVE_sum"))
    return best if best else (None, f"no PMC pass for the current {kernel}.hip (run KERNEL={kernel} scripts/pmc_traffic.sh)", None)


def conv_traffic_from_profiles():
    """HBM-side bytes per launch of vae_conv3w_kernel at its two dominant shapes, from the newest committed counter pass
    (profiles/*conv3w_traffic*.json, scripts/conv_pmc_traffic.sh) whose recorded sha256 equals vae_conv3w.hip in the tree; else None + why.
    The conv family's roofline object averages launches of MANY shapes, so there is no single per-launch figure: `traffic` stays null there and
    the per-shape numbers ride beside it."""
    import glob
    import hashlib
    src = os.path.join(ROOT, "fastvideo_amd", "csrc", "vae_conv3w.hip")
    cur = hashlib.sha256(open(src, "rb").read()).hexdigest() if os.path.exists(src) else None
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*conv3w_traffic*.json"))):
        try:
            j = json.load(open(f))
        except Exception:  # noqa: BLE001
            continue
        if j.get("kernel_source_sha256") == cur and j.get("shapes"):
            best = ({k: {a: v[a] for a in ("traffic_bytes_per_launch", "algorithmic_bytes_per_launch", "traffic_over_algorithmic",
                                           "without_fetch_correction_over_algorithmic") if a in v}
                     for k, v in j["shapes"].items()}, os.path.basename(f))
    return best if best else (None, "no counter pass for the current vae_conv3w.hip (run scripts/conv_pmc_traffic.sh)")


def vae_cpu_baseline(latent_shape, budget_frames=2):
    """CPU baseline of the VAE stage on a bounded sample: the reference's own ``AutoencoderKLWan.decode`` (fp32 on the host — the
    CPU has no bf16 autocast speed-up to offer) on the first ``budget_frames`` latent frames at the full spatial size, when a reference tree is present (live or
    staged); else the oracle port (oracle/vae_oracle.py).  Scaled to the full latent by algorithmic FLOPs."""
    from fastvideo_amd.wan_config import vae_decode_flops
    _, _, T, H, W = latent_shape
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(1)
    z = torch.randn((1, 16, budget_frames, H, W), generator=g)
    kind, where = "port", "oracle/vae_oracle.py"
    t = None
    try:
        from oracle import ref_loader as R
        if R.available():
            from oracle.make_golden_vae import build_ref_vae
            vae = build_ref_vae(base_dim=96, seed=0)
            with torch.no_grad():
                vae.decode(torch.randn((1, 16, 1, 8, 8), generator=g))  # warm-up
                t0 = time.perf_counter()
                vae.decode(z)
                t = time.perf_counter() - t0
            kind, where = "reference", "the reference's AutoencoderKLWan.decode (fastvideo/models/vaes/wanvae.py:1189-1215)"
    except Exception as ex:  # noqa: BLE001
        where += f" (reference unavailable: {repr(ex)[:120]})"
    if t is None:
        from oracle import vae_oracle as VO
        from fastvideo_amd.wan_config import wan_vae_param_spec
        sd = VO.seeded_state_dict(wan_vae_param_spec(base_dim=96), 0)
        dec = VO.WanVaeDecoderOracle(sd)
        with torch.no_grad():
            t0 = time.perf_counter()
            dec.decode(z)
            t = time.perf_counter() - t0
    scale = vae_decode_flops(T, H, W) / vae_decode_flops(budget_frames, H, W)
    per_decode = t * scale
    frames = 1 + 4 * (T - 1)
    return dict(value=round(frames / per_decode, 3), unit="pixel-frames/s", cores=threads, kind=kind,
                sample=f"{where}, torch CPU fp32, {threads} threads of {cores} cores, latent [1,16,{budget_frames},{H},{W}] in {t:.2f} s, "
                       f"scaled by algorithmic FLOPs x{scale:.2f} to {T} latent frames", ms_per_step=round(per_decode * 1e3, 1))


def measure_cfg_step(model, cfg, latent, ctx, steps=4, warmup=1):
    """The REAL denoising step of the BASELINE config ("a CFG step = 2 forwards, reported separately";
    fastvideo/pipelines/stages/denoising.py:372-596): conditional + unconditional DiT forward on the bf16 latent, then the fused
    classifier-free-guidance combine + FlowUniPC scheduler step on the fp32 latent (fvk_cfg_unipc_step) — measured both as the reference
    runs it (two forwards per step) and with the pair as ONE batch-2 forward (DenoisingLoopHip(cfg_batch=True): bit-identical per sample,
    tests/test_gpu_sched.py; half the launches, twice the grid per launch).  Dense self-attention launches are timed with HIP events in
    both modes, which answers whether the longer launches of the batched pair hold the clock better."""
    from fastvideo_amd.scheduler import DenoisingLoopHip
    dev = latent.device
    g = torch.Generator(device=dev).manual_seed(7)
    neg = torch.randn(ctx.shape, generator=g, device=dev).bfloat16()
    x0 = torch.randn(latent.shape, generator=g, device=dev)
    out = {"what": "one classifier-free-guidance denoising step = conditional + unconditional DiT forward + fused CFG combine + FlowUniPC step "
                   "(denoising.py:372-596), 50-step schedule, guidance 3.0, flow shift 3.0; same model, latent and text length as the contract line",
           "steps": steps, "warmup": warmup}
    legs = []
    for name, batch in (("two_forwards", False), ("batch2_forward", True)):
        try:
            loop = DenoisingLoopHip(model, 50, flow_shift=3.0, guidance_scale=3.0, cfg_batch=batch)
        except ValueError as e:   # per-tensor fp8 refuses the batched pair (one absmax over both samples): keep the other leg
            out[name] = {"skipped": str(e)}
            continue
        legs.append(name)
        loop.stepper.reset()
        x, x16 = x0.clone(), x0.to(torch.bfloat16)
        for i in range(warmup):
            x, x16 = loop.step(i, x, x16, ctx, neg)
        torch.cuda.synchronize()
        model.attn_events = []
        t0 = time.perf_counter()
        for i in range(warmup, warmup + steps):
            x, x16 = loop.step(i, x, x16, ctx, neg)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        events, model.attn_events = model.attn_events, None
        if not torch.isfinite(x).all():
            raise RuntimeError("cfg step: non-finite latent")
        a_ms = [a.elapsed_time(b) for a, b, *_ in events]
        fl = sum(4.0 * sq * skv * h * cfg.head_dim for _, _, sq, skv, h in events)
        out[name] = {"ms_per_step": round(ms, 3), "attn_launches_per_step": len(events) // steps,
                     "attn_mean_launch_ms": round(sum(a_ms) / len(a_ms), 4), "attn_tflops": round(fl / (sum(a_ms) * 1e-3) / 1e12, 1)}
    S = (latent.shape[2]) * (latent.shape[3] // 2) * (latent.shape[4] // 2)
    fl_step = 2 * __import__("fastvideo_amd.wan_config", fromlist=["x"]).algorithmic_flops(cfg, S, ctx.shape[1])["total"]
    for name in legs:
        out[name]["step_tflops"] = round(fl_step / (out[name]["ms_per_step"] * 1e-3) / 1e12, 1)
        out[name]["step_frac_of_bf16_peak"] = round(out[name]["step_tflops"] / PEAK_BF16_TFLOPS, 4)
    if len(legs) == 2:
        out["batch2_speedup"] = round(out["two_forwards"]["ms_per_step"] / out["batch2_forward"]["ms_per_step"], 4)
    return out


def measure_matrix_ceiling(local_rank, seconds=1.2):
    """the matrix pipe's SUSTAINED bf16 rate on THIS box at THIS power cap, measured in this process right after the
    contract measurements — a registers-only stream of v_mfma_f32_16x16x32_bf16 (the instruction of the attention, GEMM and conv kernels) on
    normal-like operands (fvk_mfma_sustained_probe_bf16; no LDS, no memory traffic, one wave per SIMD), with the socket power and shader clock
    it ran at.  It is what ANY kernel multiplying real activations could at most reach here (2049-2078 TF at 1400 W, 82 % of
    the 2.5-PF peak `roofline.frac` stays priced against); the zero-operand run beside it shows the cycle-bound rate of the same loop."""
    import importlib.util
    from fastvideo_amd import ops
    out = {"what": "registers-only v_mfma_f32_16x16x32_bf16 stream, 256 workgroups x 4 waves x 64 accumulator tiles, back to back for ~1.2 s per "
                   "operand kind, in this process after the timed region (fvk_mfma_sustained_probe_bf16)"}
    pt = None
    try:
        spec = importlib.util.spec_from_file_location("power_trace", os.path.join(ROOT, "scripts", "power_trace.py"))
        pt = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(pt)
    except Exception as ex:  # noqa: BLE001
        out["power_error"] = repr(ex)[:200]
    for name, data in (("normal_like_operands", 1), ("zero_operands", 0)):
        sampler = None
        if pt is not None:
            try:
                sampler = pt.PowerSampler(20.0, local_rank).start()
            except Exception as ex:  # noqa: BLE001
                out["power_error"] = repr(ex)[:200]
        torch.cuda.synchronize()
        w0 = time.time()
        tf, n, ms = ops.mfma_sustained_probe(seconds if data else seconds / 2, data)
        w1 = time.time()
        rec = {"tflops": round(tf, 1), "frac_of_bf16_peak": round(tf / PEAK_BF16_TFLOPS, 4), "launches": n, "ms_per_launch": round(ms, 3)}
        if sampler is not None:
            try:
                if sampler.available:
                    # skip the first 0.3 s: three warm-up launches and the clock settling under the cap
                    pw = pt.summarize(sampler.stop(), w0 + 0.3, w1, sampler.src.name)
                    rec["power_w"], rec["sclk_mhz"] = pw.get("power_w"), pw.get("sclk_mhz")
                else:
                    sampler.stop()
            except Exception as ex:  # noqa: BLE001
                rec["power_error"] = repr(ex)[:200]
        out[name] = rec
    return out


def measure_vae(latent_shape, steps, warmup, frames_per_pass=4):
    """One step = one causal-3D-conv VAE decode (``AutoencoderKLWan.decode``, fastvideo/models/vaes/wanvae.py:1189-1215) of ``latent_shape``
    on the current GPU: cfg2 [1,16,21,60,104] -> [1,3,81,480,832], cfg5 [1,16,33,90,160] -> [1,3,129,720,1280]; random-init Wan2.1-VAE
    decoder (base_dim 96, 73 M parameters).  Served precision: bf16 autocast (bf16 activations, fp32 accumulation, fp32 norms, fp32
    pixels out) — the Wan pipelines' decode default (``vae_decode_precision = "bf16"``, configs/pipelines/wan.py:59, preferred by
    decoding.py:165-167); an explicit "fp32" override is refused by WanVaeDecoderHip, not served.  Parity at real frame sizes against
    the reference's fp32 decode, bounded by the reference's own bf16-autocast error: tests/test_gpu_vae_real.py."""
    from fastvideo_amd import ops
    from fastvideo_amd.wan_config import vae_decode_flops, wan_vae_param_spec
    from fastvideo_amd.wan_vae import WanVaeDecoderHip
    g = torch.Generator().manual_seed(0)
    sd = {}
    for n, shp in wan_vae_param_spec(base_dim=96):
        fan_in = 1
        for d in shp[1:]:
            fan_in *= d
        sd[n] = (torch.ones(shp) if "gamma" in n else ((torch.rand(shp, generator=g) * 2 - 1) * (3.0 / fan_in)**0.5 if len(shp) >= 4 else torch.zeros(shp)))
    dec = WanVaeDecoderHip(sd, device="cuda", frames_per_pass=frames_per_pass, precision="bf16")
    z = torch.randn(latent_shape, generator=g).cuda()
    torch.cuda.reset_peak_memory_stats()
    for _ in range(max(warmup, 1)):
        y = dec.decode(z)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        y = dec.decode(z)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if not torch.isfinite(y).all():
        raise SystemExit("non-finite output")
    # roofline leg: per-launch HIP events around every conv launch of ONE extra decode (events inside the timed region would perturb it)
    ops.VAE_CONV_EVENTS = []
    dec.decode(z)
    torch.cuda.synchronize()
    ev, ops.VAE_CONV_EVENTS = ops.VAE_CONV_EVENTS, None
    groups = {}
    for taps, cin, cout, fl, e0, e1 in ev:
        k = "vae_conv3w_kernel + conv_out's vae_convout_kernel (3x3 spatial taps, halo slabs in LDS; one wave per SIMD, 16x16x32 MFMAs; conv_out: the three time taps on the N axis, every input frame fetched once)" if taps.endswith("3x3") else "vae_conv_kernel (1x1 / temporal taps)"
        gsum = groups.setdefault(k, [0.0, 0.0, 0])
        gsum[0] += fl
        gsum[1] += e0.elapsed_time(e1)
        gsum[2] += 1
    dom = max(groups, key=lambda k: groups[k][1])
    fl_d, ms_d, n_d = groups[dom]
    achieved = fl_d / (ms_d * 1e-3) / 1e12
    ms = elapsed / steps * 1e3
    fl = vae_decode_flops(*latent_shape[2:])
    frames = y.shape[2]
    tr_shapes, tr_src = conv_traffic_from_profiles()
    return {"metric": f"Wan2.1 VAE decode, latent {list(latent_shape)} -> pixels {list(y.shape)} (one decode per step)",
            "value": round(frames / (elapsed / steps), 2), "unit": "pixel-frames/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 (the Wan pipelines' decode default, vae_decode_precision='bf16', configs/pipelines/wan.py:59: bf16 activations / MFMA "
                     "operands, fp32 accumulation + norms, fp32 pixels; an explicit 'fp32' override is refused, not served)",
            "data": "synthetic (randn latent, random-init decoder)",
            "config": {"workload": f"Wan2.1 VAE decoder (base_dim 96), frame-chunked cached decode of latent {list(latent_shape)}", "parallelism": "1 GPU",
                       "frames_per_pass": frames_per_pass},
            "step_tflops": round(fl / (ms * 1e-3) / 1e12, 1), "step_frac_of_bf16_peak": round(fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
            "roofline": dict(bound="mfma", kernel=dom, achieved=round(achieved, 1), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                             frac=round(achieved / PEAK_BF16_TFLOPS, 4), traffic=None, traffic_by_shape=tr_shapes, traffic_source=tr_src,
                             flops_per_launch=fl_d / n_d,
                             mean_launch_ms=round(ms_d / n_d, 4), launches=n_d, share_of_step=round(ms_d / ms, 3),
                             other_kernels={k: dict(ms=round(v[1], 2), tflops=round(v[0] / (v[1] * 1e-3) / 1e12, 1), launches=v[2])
                                            for k, v in groups.items() if k != dom}),
            "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2)}


def run_vae(args):
    """--stage vae: the VAE decode of the config's latent as its own bench line (measure_vae) + its CPU baseline."""
    import __graft_entry__ as G
    G.build()
    if int(os.environ.get("WORLD_SIZE", "1")) != 1 or args.gpus != 1:
        raise SystemExit("--stage vae is a single-GPU line (tile-parallel decode is covered by tests, not benchmarked here)")
    latent_shape = {"cfg2": (1, 16, 21, 60, 104), "cfg5": (1, 16, 33, 90, 160), "cfg1": (1, 16, 9, 64, 64), "cfg4": (1, 16, 21, 90, 160)}[args.config]
    out = measure_vae(latent_shape, args.steps, args.warmup, args.vae_frames_per_pass)
    if not args.no_cpu_baseline:
        try:
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):
                out["cpu_baseline"] = vae_cpu_baseline(latent_shape)
        except Exception as ex:  # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(ex)[:300]}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--attention", default="dense", choices=["dense", "vsa", "sta"])
    ap.add_argument("--config", default="cfg2", choices=["cfg1", "cfg2", "cfg4", "cfg5"],
                    help="BASELINE.json workload: cfg2 (default, the contract line) Wan2.1-1.3B 81fx480p; cfg1 the 9x64x64 plumbing latent; "
                         "cfg4 Wan2.2-A14B 81fx720p (one expert; SP=8 in the reference config); cfg5 Wan2.1-1.3B geometry at 129fx720p")
    ap.add_argument("--quant", default=None, choices=["fp8", "fp8_channel"],
                    help="fp8 linear path (BASELINE config 5's GEMM dtype); the contract line is the default bf16 run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-power-trace", action="store_true", help="do not sample socket power / shader clock (amdsmi, 20 Hz, a host thread) "
                    "over an untimed repeat of the K steps after the timed region (the `power` object)")
    ap.add_argument("--no-cfg-step", action="store_true", help="skip the `cfg_step` sub-object (the full classifier-free-guidance denoising step: "
                    "2 forwards + fused CFG / UniPC tail, measured after the timed region)")
    ap.add_argument("--no-matrix-ceiling", action="store_true", help="skip roofline.matrix_ceiling (the registers-only MFMA stream timed after the timed region)")
    ap.add_argument("--no-vae", action="store_true", help="skip the `vae` sub-object (VAE decode of the same latent, measured after the timed region)")
    ap.add_argument("--cpu-baseline-kind", default="auto", choices=["auto", "reference", "port"],
                    help="auto: the reference itself when a reference tree is present (live or staged), else the oracle port")
    ap.add_argument("--stage", default="dit", choices=["dit", "vae"],
                    help="dit (default, the contract line): one DiT forward per step; vae: one causal-3D-conv VAE decode of the same latent per step")
    ap.add_argument("--vae-frames-per-pass", type=int, default=4,
                    help="--stage vae: latent frames per decoder pass after the first (1 = the reference's frame-by-frame walk; results are "
                         "bit-identical for every value, tests/test_gpu_vae.py)")
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer layers (result marked invalid)")
    args = ap.parse_args()
    if args.stage == "vae":
        return run_vae(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # FVK_BENCH_SHARED_GPU=1 (test hook, numbers INVALID): all ranks share cuda:0 and exchange through gloo (host-staged) — exercises the
    # multi-rank flow of this script on a one-GPU box; production = one rank per GPU over RCCL (backend "nccl")
    shared = os.environ.get("FVK_BENCH_SHARED_GPU") == "1"
    if shared:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import __graft_entry__ as G
    G.build()
    from fastvideo_amd import wan_config as WC
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip

    cfg = WC.WAN22_T2V_A14B if args.config == "cfg4" else WC.WAN21_T2V_1_3B
    if args.layers:
        cfg = WC.WanConfig(cfg.name, cfg.num_heads, cfg.head_dim, cfg.ffn_dim, args.layers)
    latent_shape = {"cfg1": WC.LATENT_CFG1, "cfg2": WC.LATENT_81F_480P, "cfg4": WC.LATENT_81F_720P, "cfg5": (1, 16, 33, 90, 160)}[args.config]
    L_text = 512
    S = (latent_shape[2] // 1) * (latent_shape[3] // 2) * (latent_shape[4] // 2)

    sd = WC.random_state_dict(cfg, seed=0, device=dev, with_vsa_gate=(args.attention == "vsa"))
    model = WanTransformer3DModelHip(sd, cfg.num_heads, cfg.head_dim, cfg.patch_size, cfg.eps, cfg.freq_dim,
                                     attention=args.attention, device=dev, quantization=args.quant, attn_autotune=True)
    del sd
    g = torch.Generator(device=dev).manual_seed(1)
    latent = torch.randn(latent_shape, generator=g, device=dev).bfloat16()
    ctx = torch.randn((1, L_text, cfg.text_dim), generator=g, device=dev).bfloat16()
    ts = torch.tensor([500.0], device=dev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up, before the W warm-up steps: the model's in-place choice between its two long-key attention kernels takes its first two forwards
    # (wan_dit.py: attn_autotune) — done here so that no warm-up or timed step carries the timing forward, whatever W is
    for _ in range(3):
        if args.attention != "dense" or not model.attn_autotune:
            break
        model(latent, ctx, ts)
    for _ in range(args.warmup):
        y = model(latent, ctx, ts)
    sync()
    model.attn_events = []
    if world > 1:
        model.sp.stats = {}  # per-exchange bytes + HIP-event pairs on the compute stream (fastvideo_amd/distributed.py)
    t0 = time.perf_counter()
    wall0 = time.time()
    for _ in range(args.steps):
        y = model(latent, ctx, ts)
    sync()
    elapsed = time.perf_counter() - t0
    wall1 = time.time()
    events, model.attn_events = model.attn_events, None
    sp_stats = None
    if world > 1:
        sp_stats, model.sp.stats = model.sp.stats or {}, None   # the exchange accounting covers the K timed steps only
    if not torch.isfinite(y.float()).all():
        raise SystemExit("non-finite output")
    # socket power / shader clock: a SEPARATE, untimed repeat of the same K steps straight after the timed region (the 20-Hz host
    # sampling thread shares the GIL with the launch loop, so it must not run inside the region `value` is computed from).  EVERY rank runs the
    # repeat (a sequence-parallel forward holds collectives: rank 0 alone would wait for its peers forever); only rank 0 samples.
    power = None
    if not args.no_power_trace:
        sampler, pt = None, None
        if rank == 0:
            try:  # scripts/power_trace.py: a host thread reading socket power + shader clock at 20 Hz (measurement only; never required)
                import importlib.util
                spec = importlib.util.spec_from_file_location("power_trace", os.path.join(ROOT, "scripts", "power_trace.py"))
                pt = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(pt)
                sampler = pt.PowerSampler(20.0, local_rank).start()
            except Exception as ex:  # noqa: BLE001
                sampler, power = None, {"error": repr(ex)[:200]}
        sync()
        tp0, pw0 = time.perf_counter(), time.time()
        for _ in range(args.steps):
            model(latent, ctx, ts)
        sync()
        pw1, tp1 = time.time(), time.perf_counter()
        if sampler is not None:
            try:
                if not sampler.available:
                    sampler.stop()
                    power = {"error": "no power / clock source readable", "sampler_errors": sampler.errors[:4]}
                else:
                    samples = sampler.stop()
                    power = pt.summarize(samples, pw0, pw1, sampler.src.name)
                    power["what"] = ("socket power and shader clock of this GPU sampled by a host thread (scripts/power_trace.py) over an UNTIMED repeat "
                                     "of the K steps straight after the timed region; a MEASURED clock, unlike roofline.clock_from_profiled_cycles_ghz")
                    power["ms_per_step_while_sampling"] = round((tp1 - tp0) * 1e3 / args.steps, 3)
                    if sampler.errors:
                        power["sampler_errors"] = sampler.errors[:4]
            except Exception as ex:  # noqa: BLE001
                power = {"error": repr(ex)[:200]}
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if shared else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # roofline of the dominant kernel (dense self-attention): algorithmic FLOPs per launch / mean launch duration
    attn_ms = [e0.elapsed_time(e1) for e0, e1, *_ in events]
    _, _, Sq, Skv, h = events[0]
    mean_ms = sum(attn_ms) / len(attn_ms)
    if args.attention == "dense":
        # launches may differ in head count (pipelined SP exchange: two head chunks per layer): total FLOPs / total time
        flops_launch = sum(4.0 * sq_ * skv_ * h_ * cfg.head_dim for _, _, sq_, skv_, h_ in events) / len(events)
        ran = model.dense_kernel_ran or "attn_w16"       # what the last self-attention launch actually ran (wan_dit._dense_attn)
        dense_kernel = "attn_w64" if ran == "attn_w64" else "attn_pp2" if ran == "attn_pp2" else "attn_w16"
        kname = (f"{dense_kernel}_kernel{ran[len(dense_kernel):]} (dense self-attention: 4 waves x 64 query rows, one wave per SIMD, "
                 f"{'32x32x16' if dense_kernel == 'attn_w64' else '16x16x32'} MFMAs, 64-key sub-tiles software-pipelined in the wave)"
                 if dense_kernel != "attn_pp2" else "attn_pp2_kernel (dense self-attention, short key axis: 8-wave ping-pong kernel)")
    else:
        # sparse modes: the algorithmic work is the selected fraction of the dense score matrix (VSA: top-k of the 64-token blocks
        # + the coarse branch, negligible; STA: the window's share of key tokens); the timed region is the whole attention
        # (coarse stage + block-sparse kernel + combine; the tile / untile gathers are folded into the neighbouring passes)
        if args.attention == "vsa":
            m = next(iter(v for k_, v in model._vsa_cache.items() if not (isinstance(k_, tuple) and k_ and isinstance(k_[0], str))))
            dens = m["topk"] / m["variable_block_sizes"].numel() * (m["S_pad"] / Skv)**2
        else:
            dens = next(iter(v for k_, v in model._vsa_cache.items() if isinstance(k_, tuple) and k_ and k_[0] == "sta"))["density"]
        flops_launch = 4.0 * Sq * Skv * h * cfg.head_dim * dens
        if args.attention == "sta":  # single GPU: no gather passes (wan_dit._sta_fused); queries packed by window class on 256-row workgroups
            kname = f"sta self-attention (V^T gather-transpose + attn_pp2_kernel over KV block lists, output rows scattered), density {dens:.3f} of dense"
        else:
            kname = (f"{args.attention} self-attention (coarse stage + attn_bs16_kernel over 64-row block lists, its last round split + merged, + combine pass; "
                     f"tile / untile folded into the neighbouring passes), density {dens:.3f} of dense")
    achieved = flops_launch / (mean_ms * 1e-3) / 1e12
    traffic, traffic_src, gui_cycles = None, None, None
    if args.attention == "dense" and args.config == "cfg2" and world == 1:
        traffic, traffic_src, gui_cycles = attention_traffic_from_profiles(dense_kernel)
    roof = dict(bound="mfma", kernel=kname, achieved=round(achieved, 1), peak=PEAK_BF16_TFLOPS,
                unit="TFLOP/s", frac=round(achieved / PEAK_BF16_TFLOPS, 4), traffic=traffic, traffic_source=traffic_src,
                flops_per_launch=flops_launch, mean_launch_ms=round(mean_ms, 4), launches=len(attn_ms),
                share_of_step=round(sum(attn_ms) / args.steps / (elapsed / args.steps * 1e3), 3))
    if args.attention == "dense" and model.attn_tune_report:
        # the model timed the two long-key kernels (same arithmetic to rounding) in place during its second warm-up forward and kept the faster
        roof["kernel_choice"] = model.attn_tune_report
    if gui_cycles:
        # DVFS separated from stalls: GRBM_GUI_ACTIVE (busy shader cycles, summed over the 8 XCDs by rocprofv3; same binary, same shape, so
        # the cycle count per launch carries over) / the LIVE launch duration = the clock this run sustained; the MFMA peak scales with it
        # NOT a per-run measurement of the clock: the cycle count comes from ONE profiled pass of this binary (a constant of the binary and
        # the shape), only the duration is live — so "FLOP per profiled busy cycle" below is the same number in every run (it is the matrix
        # pipe's busy share of the kernel's cycles, ~0.73), and the clock derived from it moves only with the live duration.  The MEASURED
        # clock and power of this run are in the `power` object (amdsmi samples over the timed region).
        clk = gui_cycles / 8 / (mean_ms * 1e-3) / 1e9
        roof.update(clock_from_profiled_cycles_ghz=round(clk, 3),
                    flop_per_profiled_cycle_frac_of_peak=round(achieved / (PEAK_BF16_TFLOPS * clk / 2.4), 4),
                    profiled_cycles_note="GRBM_GUI_ACTIVE per launch from the committed PMC pass (a constant of the binary) / the live launch "
                                         "duration; flop_per_profiled_cycle_frac_of_peak is therefore a constant of the binary, not of this run")
    fl = WC.algorithmic_flops(cfg, S, L_text)
    if args.attention != "dense":
        # sparse modes: the algorithmic work of self-attention is the SELECTED share of the score matrix, not the dense count
        fl = dict(fl)
        fl["total"] = fl["total"] - fl["self_attn"] * (1.0 - dens)
        fl["self_attn"] = fl["self_attn"] * dens
    ms_per_step = elapsed / args.steps * 1e3
    lay = model.sp.lay
    par = "sp1" if world == 1 else (f"sp{world} 2-D Ulysses (head groups {lay.G} x query blocks {lay.U}), "
                                    f"{('pipelined (2 head chunks, asynchronous exchanges' + (', two streams)' if model.sp.overlap_streams == 2 else ')')) if model.sp.overlap else 'plain'} exchange")
    out = {
        "metric": "DiT-step latent-tokens/s, Wan2.1-T2V-1.3B 81fx480p (one DiT forward per step)" if args.config == "cfg2" else
                  f"DiT-step latent-tokens/s, BASELINE {args.config} (one DiT forward per step)",
        "value": round(S / (elapsed / args.steps), 1), "unit": "latent-tokens/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "bf16" if not args.quant else f"{args.quant} linears (e4m3fn MFMA) + bf16 attention", "data": "synthetic (randn latent, random-init weights)",
        "config": {"workload": f"{cfg.name} {cfg.num_layers} layers, latent {list(latent_shape)} = {S} tokens, text 512 tokens, "
                               f"{args.attention} attention, 1 forward/step (no CFG)", "parallelism": par},
        "step_tflops": round(fl["total"] / (ms_per_step * 1e-3) / 1e12, 1),
        "step_frac_of_bf16_peak": round(fl["total"] / (ms_per_step * 1e-3) / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
        "roofline": roof,
        "power": power,
        "timed_region_unix": [round(wall0, 4), round(wall1, 4)],  # host clock around the K timed steps: scripts/power_trace.py cuts its samples here
    }
    if world > 1:
        # sequence-parallel exchange accounting of THIS rank (rank 0): bytes sent to other ranks and the time the compute stream spent
        # in each collective (HIP events bracketing it on that stream), i.e. the achieved per-GPU egress rate over xGMI
        st = sp_stats or {}
        ex = {}
        for kind, rec in st.items():
            ms = [a.elapsed_time(b) for a, b in rec["events"]]
            tot_ms = sum(ms)
            ex[kind] = dict(calls_per_step=rec["calls"] / args.steps, remote_mb_per_call=round(rec["remote_bytes"] / max(rec["calls"], 1) / 1e6, 3),
                            mean_ms=round(tot_ms / max(len(ms), 1), 4) if ms else None,
                            egress_gb_s=round(rec["remote_bytes"] / (tot_ms * 1e-3) / 1e9, 1) if tot_ms > 0 else None,
                            ms_per_step=round(tot_ms / args.steps, 3) if ms else None)
        out["exchange"] = dict(backend=dist.get_backend(), rccl_ranks=(world if dist.get_backend() == "nccl" else 0), per_kind=ex,
                               note="rank 0's view; egress = bytes to the other ranks / time the compute stream waited in the collective; "
                                    "xGMI: 7 links x ~153 GB/s per GPU (point-to-point mesh)")
    if args.layers:
        out["INVALID"] = "debug run with fewer layers"
    if shared:
        out["INVALID"] = "test hook: ranks share one GPU and exchange through gloo"
    if world == 1 and args.attention == "dense" and not args.no_cfg_step:
        # the denoising step the BASELINE config actually runs 50 times (2 forwards + the fused scheduler tail), AFTER the timed region
        try:
            out["cfg_step"] = measure_cfg_step(model, cfg, latent, ctx)
        except Exception as ex:  # noqa: BLE001 - never hide the contract number
            out["cfg_step"] = {"error": repr(ex)[:300]}
    if world == 1 and not args.no_matrix_ceiling:   # (N > 1: the other ranks would leave the process group while rank 0 still measures)
        try:
            mc = measure_matrix_ceiling(local_rank)
            roof["sustained_matrix_rate_at_cap_tf"] = mc["normal_like_operands"]["tflops"]
            roof["frac_of_sustained_matrix_rate"] = round(achieved / mc["normal_like_operands"]["tflops"], 4)
            roof["matrix_ceiling"] = mc
            out["step_frac_of_sustained_matrix_rate"] = round(out["step_tflops"] / (mc["normal_like_operands"]["tflops"] * world), 4)
        except Exception as ex:  # noqa: BLE001 - never hide the contract number
            roof["matrix_ceiling"] = {"error": repr(ex)[:300]}
    if world == 1 and args.config in ("cfg2", "cfg5") and not args.layers and not args.no_vae:
        # the other hot kernel family of the path, measured AFTER the timed region on the same latent geometry, so that the
        # driver's record carries it too: `vae` = the --stage vae line without its CPU baseline (3 decodes after 1 warm-up)
        del model, y
        torch.cuda.empty_cache()
        try:
            v = measure_vae(tuple(latent_shape), 3, 1)
            out["vae"] = {k: v[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype", "step_tflops", "step_frac_of_bf16_peak",
                                            "roofline", "peak_mem_gb", "steps", "warmup")}
        except Exception as ex:  # noqa: BLE001 - never hide the contract number
            out["vae"] = {"error": repr(ex)[:300]}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                import contextlib
                with contextlib.redirect_stdout(sys.stderr):  # the reference's logger writes to stdout: keep the ONE JSON line alone there
                    out["cpu_baseline"] = cpu_baseline(cfg, latent_shape, S, L_text, args.cpu_baseline_kind)
            except Exception as ex:  # the baseline must never hide the GPU number
                out["cpu_baseline"] = {"error": repr(ex)[:300]}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
