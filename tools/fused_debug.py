"""developer aid: compare the one-launch (XCD-local) two-pass path with torch.fft frame by frame"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chowdsp_fft_amd import api

n = 65536
dev = torch.device("cuda:0")
s = api.fft_new_setup(n, api.FFT_COMPLEX)
for batch in (1, 3, 17, 64, 257, 1024, 8192):
    g = torch.Generator(device=dev); g.manual_seed(batch)
    x = torch.rand((batch, 2 * n), device=dev, generator=g) * 2 - 1
    y = torch.empty_like(x)
    api.fft_transform_batch(s, x, y, api.FFT_FORWARD)
    torch.cuda.synchronize()
    ref = torch.view_as_real(torch.fft.fft(torch.view_as_complex(x.view(batch, n, 2).contiguous()), dim=1)).reshape(batch, 2 * n)
    err = (y - ref).norm(dim=1) / ref.norm(dim=1)
    bad = (err > 1e-5).nonzero().flatten().tolist()
    print(f"batch {batch}: max rel err {float(err.max()):.3e}, bad frames {len(bad)} {bad[:24]}")
    if bad:
        f = bad[0]
        e2 = (y[f] - ref[f]).view(n, 2).norm(dim=1)
        wrong = (e2 > 1e-3 * ref[f].abs().max()).nonzero().flatten()
        print("   frame", f, "wrong bins", wrong.numel(), "first", wrong[:16].tolist(), "kb(mod 256) of first", [int(w) % 256 for w in wrong[:16].tolist()])
