"""Soak test of the shipped attn_pp2 schedule (one barrier per step, leading-group in-stream DMA): many launches over many shapes, every
output compared bit for bit with the round-1 two-barrier schedule (attn_impl 103).  Rescale spikes at random keys make individual waves
take the slow branch (1 900 extra cycles), i.e. they skew the two wave groups against each other — the situation in which a ring slot
refilled too early would be read half-overwritten."""
import os as _os
_os.environ.setdefault("FVK_PROBE_LIB", "1")  # A/B switches exist only in the measurement build (scripts/probes/libfvk_probe.so)
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastvideo_amd import ops
random.seed(0)
g = torch.Generator(device="cuda").manual_seed(0)
bad = 0; n = 0
shapes = [(1, 3, 2048, 4096), (2, 2, 1030, 2999), (1, 12, 4096, 8190), (1, 1, 256, 16384), (1, 2, 3000, 129), (1, 4, 777, 32760)]
for it in range(int(os.environ.get("ITERS", "60"))):
    B, H, Sq, Skv = shapes[it % len(shapes)]
    q, k, v = (torch.randn((B, s_, H, 128), generator=g, device="cuda").bfloat16() for s_ in (Sq, Skv, Skv))
    for _ in range(8):  # spikes: some query rows meet a dominant key late in the sequence
        b, h = random.randrange(B), random.randrange(H)
        k[b, random.randrange(Skv // 2, Skv), h] = q[b, random.randrange(Sq), h] * 5
    ops.set_tunable("attn_impl", 103)
    ref = ops.attn_dense(q, k, v)
    ops.set_tunable("attn_impl", 0)
    for rep in range(3):
        o = ops.attn_dense(q, k, v)
        n += 1
        if not torch.equal(o, ref):
            bad += 1
            print("MISMATCH", it, rep, (B, H, Sq, Skv), (o.float() - ref.float()).abs().max().item(), flush=True)
torch.cuda.synchronize()
print(f"launches {n} mismatches {bad}")
sys.exit(1 if bad else 0)
