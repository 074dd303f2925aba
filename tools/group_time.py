"""developer aid: N = 65536 complex, the three two-pass forms side by side (two launches / ticket queues / static groups):
parity of each against the two-launch result on a few frames, then HIP-event time per direction.
usage: group_time.py [batch] [forms, e.g. 0,1,2]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chowdsp_fft_amd import api

n = 65536
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
forms = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "1", "2"])]
dev = torch.device("cuda:0")
s = api.fft_new_setup(n, api.FFT_COMPLEX)
x = torch.rand((batch, 2 * n), device=dev) * 2 - 1
y = torch.empty_like(x)
api.set_option("fused_min_batch", 1)
api.set_option("fused_two_pass", 0)
api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
torch.cuda.synchronize()
pick = sorted(set([0, 1, batch // 2, batch - 1]))
want = y[pick].clone()
names = {0: "two launches", 1: "ticket queues", 2: "static groups"}
for form in forms:
    api.set_option("fused_two_pass", form)
    y.zero_()
    api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
    api.stream_synchronize()
    err = float(((y[pick] - want).norm(dim=1) / want.norm(dim=1)).max())
    for direction, dn in ((api.FFT_FORWARD, "fwd"), (api.FFT_BACKWARD, "inv")):
        for _ in range(3):
            api.fft_transform_batch(s, x, y, direction)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            api.fft_transform_batch(s, x, y, direction)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"form {form} ({names[form]:13s}) {dn}: {ms:.3f} ms  {batch * 2 * n * 8 / ms / 1e6:7.0f} GB/s algorithmic = {batch * 2 * n * 8 / ms / 1e6 / 8000:.3f} of 8 TB/s   rel-L2 vs two-launch {err:.2e}", flush=True)
api.set_option("fused_two_pass", 0)
