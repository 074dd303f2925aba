"""developer aid: correctness + rate of a batched complex transform (HIP events), for A/B runs under different env switches"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chowdsp_fft_amd import api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dev = torch.device("cuda:0")
s = api.fft_new_setup(n, api.FFT_COMPLEX)
x = torch.rand((batch, 2 * n), device=dev) * 2 - 1
y = torch.empty_like(x); z = torch.empty_like(x)
api.fft_transform_batch(s, x, y, api.FFT_FORWARD); api.fft_transform_batch(s, y, z, api.FFT_BACKWARD)
torch.cuda.synchronize()
ref = torch.view_as_real(torch.fft.fft(torch.view_as_complex(x[:64].view(64, n, 2).contiguous()), dim=1)).reshape(64, -1)
err = float(((y[:64] - ref).norm(dim=1) / ref.norm(dim=1)).max())
rt = float((z / n - x).abs().max())
for _ in range(20):
    api.fft_transform_batch(s, x, y, api.FFT_FORWARD); api.fft_transform_batch(s, y, z, api.FFT_BACKWARD)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    api.fft_transform_batch(s, x, y, api.FFT_FORWARD); api.fft_transform_batch(s, y, z, api.FFT_BACKWARD)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / (2 * reps)
print(f"N={n} batch={batch}: fwd rel err vs torch.fft {err:.2e}, round trip max abs {rt:.2e}; {ms:.4f} ms per launch, {batch * n * 16 / ms / 1e6:.0f} GB/s")
