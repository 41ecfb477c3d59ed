"""tests/golden/fp8_quant.pt from the REAL reference quantisers (fastvideo/layers/quantization/fp8_config.py), CPU."""
import os, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_loader as R
R.install()
from fastvideo.layers.quantization import fp8_config as F8
g = torch.Generator().manual_seed(0)
x = (torch.randn((37, 64), generator=g) * torch.logspace(-3, 2, 37).unsqueeze(1)).bfloat16()   # rows spanning 5 decades
x[5] = 0                                                                                      # an all-zero row (scale clamp)
qt, st = F8._quantize_tensorwise(x)
qr, sr = F8._quantize_rowwise(x)
out = os.path.join(os.path.dirname(HERE), "tests", "golden", "fp8_quant.pt")
torch.save({"x": x, "q_tensor": qt.view(torch.uint8), "s_tensor": st, "q_row": qr.view(torch.uint8), "s_row": sr}, out)
print("wrote", out, os.path.getsize(out))
