"""The backward transform of the long real power-of-two frames with 1024-point rows (N = 2^19, 2^20) at batch sizes that
exercise the row sweep's work split: one frame, fewer tiles than the resident grid, more, and a chunked call (more frames than
one 128 MiB workspace holds: 2^19 -> 57 frames, 2^20 -> 29 per chunk).  Written with the carried-element variant of that sweep
(profiles/r05_real4_carry_experiment.txt: measured, not adopted); the cases hold for whichever kernel the library picks.

Backward parity against the oracle on spectra the oracle made (reference: pffft_transform_ordered BACKWARD,
/root/reference/chowdsp_fft.h:117-134), the forward . backward round trip N x, and the X0 / X_{N/2} slot.
"""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import assert_frames_close
from oracle_lib import BACKWARD, FORWARD, REAL

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,batch", [(1 << 19, 1), (1 << 19, 16), (1 << 19, 67), (1 << 20, 1), (1 << 20, 8), (1 << 20, 35)])
def test_backward_long_rows_vs_oracle(gpu, oracle, n, batch):
    api, torch, dev = gpu
    x = ol.uniform(batch * n, 900 + batch).reshape(batch, n)
    s, so = api.fft_new_setup(n, api.FFT_REAL), oracle.new_setup(n, REAL)
    assert s is not None
    spec = oracle.transform_batch(so, x, FORWARD)
    want = oracle.transform_batch(so, spec, BACKWARD)
    sd = torch.from_numpy(spec).to(dev)
    yd = torch.empty_like(sd)
    api.fft_transform_batch(s, sd, yd, api.FFT_BACKWARD)
    torch.cuda.synchronize()
    got = yd.cpu().numpy()
    assert_frames_close(got, want, what=f"real bwd N={n} batch={batch}")
    assert_frames_close(got / n, x, what=f"round trip N={n} batch={batch}")
    # in place, and the X0 / X_{N/2} slot (its two floats are two different bins: the tile at kb0 = 0 handles both)
    probe = np.zeros((2, n), dtype=np.float32)
    probe[0, 0] = 1.0  # X0 alone: a constant
    probe[1, 1] = 1.0  # X_{N/2} alone: (-1)^t
    pd = torch.from_numpy(probe).to(dev)
    api.fft_transform_batch(s, pd, pd, api.FFT_BACKWARD)
    torch.cuda.synchronize()
    p = pd.cpu().numpy()
    np.testing.assert_allclose(p[0], 1.0, atol=1e-6)
    np.testing.assert_allclose(p[1], np.where(np.arange(n) % 2 == 0, 1.0, -1.0), atol=1e-6)
    api.fft_destroy_setup(s)
    oracle.destroy_setup(so)
