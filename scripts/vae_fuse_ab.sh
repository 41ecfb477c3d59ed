cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python -m pytest tests/test_gpu_vae.py tests/test_gpu_vae_tiled.py -x -q 2>&1 | tail -2
python - <<'PY'
import json, time, torch, sys
sys.path.insert(0, ".")
from fastvideo_amd.wan_config import wan_vae_param_spec
from fastvideo_amd.wan_vae import WanVaeDecoderHip
g = torch.Generator().manual_seed(0)
sd = {}
for n, s in wan_vae_param_spec(base_dim=96):
    fan_in = 1
    for d in s[1:]: fan_in *= d
    sd[n] = torch.ones(s) if "gamma" in n else (((torch.rand(s, generator=g) * 2 - 1) * (3.0 / fan_in)**0.5) if len(s) >= 4 else torch.zeros(s))
z = torch.randn((1, 16, 21, 60, 104), generator=g).cuda()
decs = {}
for name, fc in (("fuse 96+192", (96, 192)), ("fuse 96 only", (96,))):
    d = WanVaeDecoderHip(sd); d.FUSE_NORM_C = fc; decs[name] = d
res = {k: [] for k in decs}
for r in range(4):
    for k, d in decs.items():
        d.decode(z); torch.cuda.synchronize()
        t0 = time.perf_counter(); d.decode(z); d.decode(z); torch.cuda.synchronize()
        res[k].append(round((time.perf_counter() - t0) / 2 * 1e3, 2))
print(json.dumps(res))
PY
