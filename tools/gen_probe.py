import sys, numpy as np, torch
sys.path.insert(0, ".")
from chowdsp_fft_amd import api
dev = torch.device("cuda:0")
for n in (400, 800, 1600, 3200, 6400, 2000, 4000, 8000, 1200, 2400, 4800, 9216, 4608, 2304, 1152, 576, 288, 144, 720, 360*16):
    if not api.size_supported(n, 1): print(n, "unsupported"); continue
    s = api.fft_new_setup(n, 1)
    x = np.random.default_rng(n).uniform(-1,1,(3,2*n)).astype(np.float32)
    xd = torch.from_numpy(x).to(dev); yd = torch.empty_like(xd)
    api.fft_transform_batch(s, xd, yd, 0); torch.cuda.synchronize()
    y = yd.cpu().numpy().astype(np.float64); got = y[:,0::2]+1j*y[:,1::2]
    ref = np.fft.fft(x[:,0::2].astype(np.float64)+1j*x[:,1::2], axis=1)
    print(n, api.plan_describe(n,1)["radices"], "relL2 %.2e" % (np.linalg.norm(got-ref)/np.linalg.norm(ref)))
