import os
os.environ["FVK_PROBE_LIB"]="1"
import sys; sys.path.insert(0,"/root/repo")
import torch
from fastvideo_amd import ops
torch.manual_seed(0)
M,N,K=513,256,4096
x=torch.randn(M,K).bfloat16().cuda(); w=(torch.randn(N,K)*K**-0.5).bfloat16().cuda(); b=torch.randn(N).bfloat16().cuda()
res=(torch.randn(M,N)*2).bfloat16().cuda(); gate=(torch.randn(3,N)*0.5).cuda()
def run(impl):
    ops.set_tunable("gemm_impl", impl)
    o=ops.gemm(x,w,b,epilogue=ops.EPI_RESIDUAL_GATE,residual=res,gate=gate).cpu()
    ops.set_tunable("gemm_impl", 0); return o
seq=[125,125,0,0,61,125,0,29,0,29,125]
outs=[run(i) for i in seq]
ref=outs[0]
for i,o in zip(seq,outs):
    d=(o.float()-ref.float()).abs()
    print(i, "equal" if torch.equal(o,ref) else f"DIFF n={int((d>0).sum())} max={d.max().item():.4g} rows={sorted(set((d>0).nonzero()[:,0].tolist()))[:12]}")
