"""Which vendor-library kernels torch.matmul picks at the DiT GEMM shapes (run under rocprofv3 --kernel-trace --stats)."""
import torch
S, d, F = 32760, 1536, 8960
for M, N, K in ((S, 3 * d, d), (S, d, d), (S, F, d), (S, d, F), (8192, 8192, 8192), (75600, 15360, 5120), (75600, 5120, 13824)):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): torch.matmul(a, w.t(), out=out)
    torch.cuda.synchronize()
