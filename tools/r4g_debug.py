import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from chowdsp_fft_amd import api
api.set_option("real4_generic", 1)
dev = torch.device("cuda:0")
for n in (61440, 76800, 122880, 153600, 245760):
    d = api.plan_describe(n, api.FFT_REAL)
    s = api.fft_new_setup(n, api.FFT_REAL)
    x = (np.random.default_rng(1).random((3, n), dtype=np.float32) * 2 - 1)
    xd = torch.from_numpy(x).to(dev); yd = torch.empty_like(xd)
    api.fft_transform_batch(s, xd, yd, api.FFT_FORWARD); torch.cuda.synchronize()
    t = np.fft.rfft(x.astype(np.float64), axis=1)
    want = np.empty((3, n)); want[:, 0], want[:, 1] = t[:, 0].real, t[:, n // 2].real
    want[:, 2::2], want[:, 3::2] = t[:, 1:n // 2].real, t[:, 1:n // 2].imag
    got = yd.cpu().numpy().astype(np.float64)
    ef = np.linalg.norm(got - want) / np.linalg.norm(want)
    bad = np.argwhere(~np.isfinite(got) | (np.abs(got - want) > 1e-3 * np.abs(want).max()))
    spec = torch.from_numpy(want.astype(np.float32)).to(dev); zd = torch.empty_like(spec)
    api.fft_transform_batch(s, spec, zd, api.FFT_BACKWARD); torch.cuda.synchronize()
    eb = float(np.linalg.norm(zd.cpu().numpy() / n - x) / np.linalg.norm(x))
    print(n, d["n1"], d["n2"], "fwd rel", ef, "bad", len(bad), bad[:6].tolist(), "bwd rel", eb, flush=True)
