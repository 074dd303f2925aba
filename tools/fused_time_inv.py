"""developer aid: like fused_time.py, backward direction"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chowdsp_fft_amd import api
n = int(sys.argv[1]); batch = int(sys.argv[2]); real = len(sys.argv) > 3 and sys.argv[3] == "real"
dev = torch.device("cuda:0")
s = api.fft_new_setup(n, api.FFT_REAL if real else api.FFT_COMPLEX)
nf = n if real else 2 * n
x = torch.rand((batch, nf), device=dev) * 2 - 1
y = torch.empty_like(x)
for _ in range(3):
    api.fft_transform_batch(s, x, y, api.FFT_BACKWARD)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    api.fft_transform_batch(s, x, y, api.FFT_BACKWARD)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"N={n} batch={batch} {'real' if real else 'complex'} backward: {ms:.3f} ms  {batch * nf * 8 / ms / 1e6:.0f} GB/s algorithmic")
