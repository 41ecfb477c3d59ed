"""RCCL (torch.distributed backend "nccl" on ROCm) executed on the MI355X with a ONE-rank group: the GPU box has a single GPU, so the
multi-rank exchanges cannot run there (they are covered bit-exactly by the gloo world-2/4/8 tests and by multi-process runs sharing the
GPU) — but the library path the N > 1 bench takes (process-group creation on a device, all_to_all_single / all_gather_into_tensor /
all_reduce on DEVICE tensors, the Mi355xCommunicator wrappers and SequenceParallel's exchange helpers) is executed here end to end.
Runs in a subprocess so that the default process group of the pytest process stays untouched."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
from fastvideo_amd.distributed import Mi355xCommunicator, SequenceParallel
x = torch.arange(4 * 6 * 8, dtype=torch.float32, device=dev).view(4, 6, 8).bfloat16()
recv = torch.empty_like(x)
dist.all_to_all_single(recv, x)                       # RCCL all-to-all on device memory
assert torch.equal(recv, x)
out = torch.empty_like(x)
dist.all_gather_into_tensor(out, x)
assert torch.equal(out, x)
y = x.float().clone(); dist.all_reduce(y); assert torch.equal(y, x.float())
sp = SequenceParallel(12)
assert sp.lay.P == 1 and not sp._stage_host        # "nccl": device pointers go straight to the collective
sp.stats = {{}}
r = sp._a2a(x.reshape(24, 8), None, None, 24)          # the helper every exchange goes through
assert torch.equal(r, x.reshape(24, 8))
qkv3 = torch.arange(3 * 4 * 12 * 4, dtype=torch.float32, device=dev).view(3, 4, 12, 4).bfloat16()   # q, k, v [Sl=4, H=12, D=4]
send = sp.pack_rows(qkv3[0], qkv3[1], qkv3[2])
rx = sp.exchange_rows(send)
q_blk, k_all, v_all = sp.views_of(rx, 4)
assert k_all.shape == (4, 12, 4) and "exchange1" in sp.stats and sp.stats["exchange1"]["calls"] == 1
comm = Mi355xCommunicator(dist.group.WORLD, device=dev)
assert torch.equal(comm.all_reduce(x.float()), x.float()) and torch.equal(comm.all_gather(x, dim=1), x)
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK")
"""


def test_rccl_single_rank_paths():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _SCRIPT.format(root=ROOT)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=300)
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, r.stdout[-3000:]
