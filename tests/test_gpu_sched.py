"""CFG combine + FlowUniPC step on the GPU (fvk_cfg_unipc_step) vs the REAL reference scheduler's outputs (tests/golden/unipc.pt, made
by oracle/make_golden_sched.py) and vs the oracle for other schedules.
Tolerance: the tensor arithmetic is bit-exact (fp32, reference operation order, no contraction) GIVEN the step's scalar coefficients;
the order-2 corrector's rhos come from torch.linalg.solve on the HOST (as in the reference, :595), whose last bit depends on the
host CPU's LAPACK code path — so against a fixture generated on another machine we allow 4 ulp (5e-6 abs at |x| <= 8), and demand exact
equality against the oracle run on the same host."""
import os
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "unipc.pt")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def test_unipc_cfg_matches_reference_golden_bit_exact():
    _need_gpu()
    from fastvideo_amd.scheduler import FlowUniPCStepper
    g = torch.load(GOLD)
    st = FlowUniPCStepper(g["steps"], shift=g["shift"])
    assert torch.equal(st.timesteps, g["timesteps"]) and torch.equal(st.sigmas, g["sigmas"])
    x = g["latents0"].cuda()
    for i in range(g["steps"]):
        x, x16 = st.step(g["text"][i].cuda(), x, g["uncond"][i].cuda(), g["guidance"])
        d = (x.cpu() - g["latents"][i]).abs().max().item()
        assert d <= 5e-6, f"step {i}: max diff {d}"
        if i < 2:  # no linalg.solve involved yet: exact
            assert torch.equal(x.cpu(), g["latents"][i]) and torch.equal(x16.cpu(), g["latents"][i].bfloat16())
        x = g["latents"][i].cuda()
    with pytest.raises(RuntimeError):
        st.step(g["text"][0].cuda(), x)   # past the end of the schedule


@pytest.mark.parametrize("steps,shift,order", [(3, 8.0, 2), (50, 3.0, 2), (4, 5.0, 1), (1, 3.0, 2)])
def test_unipc_no_guidance_matches_oracle(steps, shift, order):
    _need_gpu()
    from fastvideo_amd.scheduler import FlowUniPCStepper
    from oracle.sched_oracle import FlowUniPCOracle
    o, st = FlowUniPCOracle(steps, shift=shift, solver_order=order), FlowUniPCStepper(steps, shift=shift, solver_order=order)
    gen = torch.Generator().manual_seed(steps)
    x = torch.randn((1, 16, 2, 9, 7), generator=gen)   # ragged element count (2016, not a multiple of 1024)
    xg = x.cuda()
    for _ in range(steps):
        mo = torch.randn(x.shape, generator=gen).bfloat16()
        x = o.step(mo, x)
        xg, _ = st.step(mo.cuda(), xg, want_bf16=False)
        assert torch.equal(xg.cpu(), x)


@pytest.mark.parametrize("steps,shift,g", [(6, 3.0, 5.0), (4, 8.0, 3.0), (3, 5.0, None)])
def test_fp32_scalar_mode_is_torch_gpu_eager_bit_exact(steps, shift, g):
    """``scalar_rounding="fp32"`` (DenoisingLoopHip's default) claims to be what the reference's eager ops do ON AN ACCELERATOR: a 0-d
    CPU fp32 ``sigma_t`` / a Python ``guidance_scale`` meeting a bf16 device tensor stays fp32 inside the elementwise kernel (on the
    CPU the scalar is first cast to bf16 — ``scalar_rounding="bf16"``, pinned to tests/golden/unipc.pt above).  Checked here by
    running the SAME restatement (oracle/sched_oracle.py, pinned bit-exact to the real scheduler class on the CPU) as torch eager ops
    on the device — bf16 device tensors, 0-d CPU fp32 coefficients, exactly the operand types the reference's step() sees on a GPU —
    against the fused step kernel in "fp32" mode: bit-exact fp32 latents and bf16 model inputs at every step."""
    _need_gpu()
    from fastvideo_amd.scheduler import FlowUniPCStepper
    from oracle.sched_oracle import FlowUniPCOracle, cfg_combine
    o = FlowUniPCOracle(steps, shift=shift)
    st = FlowUniPCStepper(steps, shift=shift, scalar_rounding="fp32")
    st16 = FlowUniPCStepper(steps, shift=shift, scalar_rounding="bf16")
    gen = torch.Generator().manual_seed(steps)
    x = torch.randn((1, 16, 3, 9, 7), generator=gen).cuda()
    xk, xb = x.clone(), x.clone()
    differs = False
    for i in range(steps):
        text = (torch.randn(x.shape, generator=gen) * 1.7).bfloat16().cuda()
        unc = None if g is None else (torch.randn(x.shape, generator=gen) * 1.7).bfloat16().cuda()
        x = o.step(cfg_combine(text, unc, g), x)                 # torch eager ON THE DEVICE
        xk, x16 = st.step(text, xk, unc, g if g is not None else 1.0)
        assert x.is_cuda and torch.equal(xk, x), f"step {i}: max diff {(xk - x).abs().max().item()}"
        assert torch.equal(x16, x.bfloat16())
        xb, _ = st16.step(text, xb, unc, g if g is not None else 1.0)
        differs = differs or not torch.equal(xb, xk)
        xb = xk.clone()
    assert differs, "the two scalar modes never differed: the test would not notice a wrong default"


def test_denoising_loop_two_steps_vs_oracle(golden_dir):
    """Two CFG denoising steps of the tiny Wan model: HIP DiT + fused step vs oracle DiT + oracle scheduler (reference loop
    fastvideo/pipelines/stages/denoising.py:372-596).  Tolerance: the DiT's (atol 1e-1, rtol 1e-2 on bf16 outputs) carried through
    two scheduler steps of O(1) coefficients; we assert a mean error well inside it."""
    _need_gpu()
    from fastvideo_amd.scheduler import DenoisingLoopHip
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    from oracle import wan_oracle as W
    from oracle.sched_oracle import FlowUniPCOracle, cfg_combine
    fx = torch.load(os.path.join(golden_dir, "wan_tiny.pt"), weights_only=False)
    H = fx["config"]["num_heads"]
    c = fx["cases"][0]
    gen = torch.Generator().manual_seed(2)
    lat = torch.randn(c["latent"].shape, generator=gen)
    neg = torch.randn(c["ctx"].shape, generator=gen).bfloat16()
    orc, sch = W.WanOracle(fx["state_dict"], num_heads=H), FlowUniPCOracle(4, shift=3.0)
    x = lat.clone()
    with torch.no_grad():
        for i in range(2):
            t = sch.timesteps[i].float().reshape(1)
            x16 = x.bfloat16()
            x = sch.step(cfg_combine(orc.forward(x16, c["ctx"], t), orc.forward(x16, neg, t), 4.0), x)
    model = WanTransformer3DModelHip(fx["state_dict"], num_heads=H)
    y = DenoisingLoopHip(model, 4, flow_shift=3.0, guidance_scale=4.0).run(lat.cuda(), c["ctx"].cuda(), neg.cuda(), num_steps=2).cpu()
    err = (y - x).abs()
    print(f"denoising loop: max|err|={err.max().item():.4g} mean|err|={err.mean().item():.4g} ref_absmean={x.abs().mean().item():.4g}")
    assert err.mean().item() < 2e-2 and err.max().item() < 0.5
    # cfg_batch: the conditional / unconditional pair as ONE batch-2 forward == the two forwards, bit for bit (dense, one GPU: the batch is
    # a grid dimension of every kernel, rows and batch elements are independent), through two full steps
    yb = DenoisingLoopHip(model, 4, flow_shift=3.0, guidance_scale=4.0, cfg_batch=True).run(lat.cuda(), c["ctx"].cuda(), neg.cuda(), num_steps=2).cpu()
    assert torch.equal(yb, y), f"cfg_batch differs from two forwards: max {(yb - y).abs().max().item()}"


@pytest.mark.parametrize("attention", ["dense", "vsa", "sta"])
def test_batch2_forward_equals_two_forwards(golden_dir, attention):
    """WanTransformer3DModelHip on a batch of two (same latent, two prompts, the CFG pair) == the two single forwards bit for bit, in every
    attention mode (dense: one attention launch for the batch; sparse modes: per-sample launches)."""
    _need_gpu()
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    fx = torch.load(os.path.join(golden_dir, "wan_tiny.pt"), weights_only=False)
    H = fx["config"]["num_heads"]
    kw = {} if attention == "dense" else dict(attention="vsa", vsa_sparsity=0.5) if attention == "vsa" else dict(attention="sta", sta_window=(1, 3, 1), sta_tile=(2, 4, 8))
    model = WanTransformer3DModelHip(fx["state_dict"], num_heads=H, **kw)
    gen = torch.Generator().manual_seed(3)
    lat = torch.randn((1, 16, 5, 20, 36), generator=gen).bfloat16().cuda()   # 5 x 10 x 18 = 900 tokens: long enough for the 256-row kernels
    c = fx["cases"][0]
    neg = torch.randn(c["ctx"].shape, generator=gen).bfloat16().cuda()
    t = torch.tensor([617.0]).cuda()
    a, b = model(lat, c["ctx"].cuda(), t), model(lat, neg, t)
    pair = model(lat.expand(2, -1, -1, -1, -1).contiguous(), torch.cat([c["ctx"].cuda(), neg], 0), t.repeat(2))
    assert torch.equal(pair[0:1], a) and torch.equal(pair[1:2], b)
    assert not torch.equal(a, b)


def test_batch2_forward_with_fp8_linears(golden_dir):
    """per-token fp8 (fp8_channel) keeps the batch-2 == two-forwards identity (every activation row carries its own scale); per-tensor
    fp8 does NOT (one absmax over the whole [B*S, d] matrix: the pair shares a scale) — DenoisingLoopHip refuses cfg_batch for it instead of
    silently changing the numbers."""
    _need_gpu()
    from fastvideo_amd.scheduler import DenoisingLoopHip
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    fx = torch.load(os.path.join(golden_dir, "wan_tiny.pt"), weights_only=False)
    H = fx["config"]["num_heads"]
    gen = torch.Generator().manual_seed(3)
    lat = torch.randn((1, 16, 5, 20, 36), generator=gen).bfloat16().cuda()
    c = fx["cases"][0]
    neg = (torch.randn(c["ctx"].shape, generator=gen) * 3).bfloat16().cuda()      # a louder negative prompt: the two samples' absmax differ
    t = torch.tensor([617.0]).cuda()
    model = WanTransformer3DModelHip(fx["state_dict"], num_heads=H, quantization="fp8_channel")
    a, b = model(lat, c["ctx"].cuda(), t), model(lat, neg, t)
    pair = model(lat.expand(2, -1, -1, -1, -1).contiguous(), torch.cat([c["ctx"].cuda(), neg], 0), t.repeat(2))
    assert torch.equal(pair[0:1], a) and torch.equal(pair[1:2], b) and not torch.equal(a, b)
    DenoisingLoopHip(model, 4, cfg_batch=True)
    model_t = WanTransformer3DModelHip(fx["state_dict"], num_heads=H, quantization="fp8")
    with pytest.raises(ValueError, match="per-tensor fp8"):
        DenoisingLoopHip(model_t, 4, cfg_batch=True)
    DenoisingLoopHip(model_t, 4, cfg_batch=False)


def test_two_expert_loop_switches_at_the_boundary(golden_dir):
    """Wan2.2-A14B style loop: steps with t >= boundary_ratio * 1000 use the high-noise expert and guidance_scale, the others the
    low-noise expert and guidance_scale_2 (denoising.py:251-256, 377-403).  4 steps, shift 3 => t = 999, 899, 749, 499; boundary 0.875
    => expert 1 for the first two steps, expert 2 afterwards.  vs the oracle DiTs + oracle scheduler driven the same way."""
    _need_gpu()
    from fastvideo_amd.scheduler import DenoisingLoopHip
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    from oracle import wan_oracle as W
    from oracle.sched_oracle import FlowUniPCOracle, cfg_combine
    fx = torch.load(os.path.join(golden_dir, "wan_tiny.pt"), weights_only=False)
    H = fx["config"]["num_heads"]
    c = fx["cases"][0]
    sd2 = {k: v.clone() for k, v in fx["state_dict"].items()}
    sd2["proj_out.weight"] = -0.5 * sd2["proj_out.weight"]          # a visibly different second expert
    gen = torch.Generator().manual_seed(3)
    lat = torch.randn(c["latent"].shape, generator=gen)
    neg = torch.randn(c["ctx"].shape, generator=gen).bfloat16()
    o1, o2 = W.WanOracle(fx["state_dict"], num_heads=H), W.WanOracle(sd2, num_heads=H)
    sch = FlowUniPCOracle(4, shift=3.0)
    x = lat.clone()
    used = []
    with torch.no_grad():
        for i in range(3):
            t = sch.timesteps[i].float().reshape(1)
            orc, g = (o1, 4.0) if float(sch.timesteps[i]) >= 875.0 else (o2, 3.0)
            used.append(1 if orc is o1 else 2)
            x16 = x.bfloat16()
            x = sch.step(cfg_combine(orc.forward(x16, c["ctx"], t), orc.forward(x16, neg, t), g), x)
    assert used == [1, 1, 2]
    m1, m2 = WanTransformer3DModelHip(fx["state_dict"], num_heads=H), WanTransformer3DModelHip(sd2, num_heads=H)
    loop = DenoisingLoopHip(m1, 4, flow_shift=3.0, guidance_scale=4.0, transformer_2=m2, boundary_ratio=0.875, guidance_scale_2=3.0)
    assert loop.expert_for(899.0) == (m1, 4.0) and loop.expert_for(749.0) == (m2, 3.0) and loop.expert_for(875.0)[0] is m1
    y = loop.run(lat.cuda(), c["ctx"].cuda(), neg.cuda(), num_steps=3).cpu()
    err = (y - x).abs()
    print(f"two-expert loop: max|err|={err.max().item():.4g} mean|err|={err.mean().item():.4g}")
    assert err.mean().item() < 3e-2 and err.max().item() < 0.75
    # and the switch matters: the single-expert loop lands somewhere else
    y1 = DenoisingLoopHip(m1, 4, flow_shift=3.0, guidance_scale=4.0).run(lat.cuda(), c["ctx"].cuda(), neg.cuda(), num_steps=3).cpu()
    assert (y1 - x).abs().mean().item() > 4 * err.mean().item()


def test_dmd_step_bit_exact_vs_reference(golden_dir):
    """fvk_dmd_step vs the REAL pred_noise_to_pred_video / FlowMatchEulerDiscreteScheduler.add_noise outputs (tests/golden/dmd.pt)."""
    from fastvideo_amd.scheduler import DmdStepper
    fx = torch.load(os.path.join(golden_dir, "dmd.pt"), weights_only=False)
    st = DmdStepper(fx["shift"])
    for c in fx["cases"]:
        v, n = st.step(c["pred"].cuda(), c["noisy"].cuda(), c["t"], None if c["noise"] is None else c["noise"].cuda(), c["t_next"])
        assert v.dtype == torch.bfloat16 and torch.equal(v.cpu(), c["video"])
        if c["noise"] is None:
            assert n is None
        else:
            assert torch.equal(n.cpu(), c["next"])
    with pytest.raises(RuntimeError):
        st.step(fx["cases"][0]["pred"], fx["cases"][0]["noisy"], fx["cases"][0]["t"])


def test_dmd_loop_matches_oracle_loop(golden_dir):
    """DmdDenoisingLoopHip (the FastWan sampling loop, denoising.py:1322-1401) vs the same loop on the oracle DiT with the oracle DMD arithmetic
    and the SAME re-noising draws."""
    from fastvideo_amd.scheduler import DmdDenoisingLoopHip
    from fastvideo_amd.wan_dit import WanTransformer3DModelHip
    from oracle import dmd_oracle as D
    from oracle import wan_oracle as W
    fx = torch.load(os.path.join(golden_dir, "wan_tiny.pt"), weights_only=False)
    H = fx["config"]["num_heads"]
    c = fx["cases"][0]
    g = torch.Generator().manual_seed(9)
    lat = torch.randn(c["latent"].shape, generator=g)
    steps = [1000, 757, 522]
    B, C, T, Hh, Ww = lat.shape
    draws = [torch.randn(1, T, C, Hh, Ww, generator=g).bfloat16() for _ in range(2)]
    orc = W.WanOracle(fx["state_dict"], num_heads=H)
    with torch.no_grad():
        ref = D.dmd_rollout(lambda x, t: orc.forward(x, c["ctx"], t.float()), lat, steps, draws)
    it = iter(draws)
    y = DmdDenoisingLoopHip(WanTransformer3DModelHip(fx["state_dict"], num_heads=H), steps).run(lat.cuda(), c["ctx"].cuda(), lambda s_, d_: next(it))
    assert y.shape == lat.shape and y.dtype == torch.bfloat16 and next(it, None) is None
    err = (y.float().cpu() - ref.float()).abs()
    print(f"DMD loop: max|err|={err.max().item():.4g} mean|err|={err.mean().item():.4g}")
    assert err.mean().item() < 2e-2 and err.max().item() < 0.5
