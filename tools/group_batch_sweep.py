"""developer aid: N = 65536 complex forward, two launches (form 0) vs the static-group one-launch form (2) over the batch:
where does the one-launch form start to pay?  (its launch registers the whole grid first)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chowdsp_fft_amd import api

n = 65536
dev = torch.device("cuda:0")
s = api.fft_new_setup(n, api.FFT_COMPLEX)
api.set_option("fused_min_batch", 1)
for batch in (1, 8, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192):
    x = torch.rand((batch, 2 * n), device=dev) * 2 - 1
    y = torch.empty_like(x)
    row = []
    for form in (0, 2):
        api.set_option("fused_two_pass", form)
        for _ in range(3):
            api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
        torch.cuda.synchronize()
        reps = 20 if batch >= 1024 else 100
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / reps * 1e3)
    print(f"batch {batch:5d}: two launches {row[0]:9.1f} us   static groups {row[1]:9.1f} us   ratio {row[1] / row[0]:.3f}", flush=True)
api.set_option("fused_two_pass", 0)
