"""SURVEY section 8 row f4, 2-D half: fft_transform_2d_batch = the oracle's 1-D transform applied along each axis
(rows first, then columns), and the float64 2-D transform.  Row lengths cover the on-chip, split-plane, two-pass and
mixed-radix plans; row counts cover every tile width (16 / 8 / 4 columns), ragged column tiles and mixed radices."""
import numpy as np
import pytest

import oracle_lib as ol
from conftest import assert_frames_close
from oracle_lib import BACKWARD, COMPLEX, FORWARD

pytestmark = pytest.mark.gpu


def _oracle_2d(oracle, x, rows, cols, direction):
    """x: [batch, rows, 2*cols] float32 -> the oracle's transform along the rows, then along the columns"""
    batch = x.shape[0]
    sc = oracle.new_setup(cols, COMPLEX)
    y = oracle.transform_batch(sc, x.reshape(batch * rows, 2 * cols), direction).reshape(batch, rows, cols, 2)
    oracle.destroy_setup(sc)
    sr = oracle.new_setup(rows, COMPLEX)
    if sr is None:  # the reference needs rows % 16 == 0; other row counts are checked against float64 only
        return None
    colmajor = np.ascontiguousarray(y.transpose(0, 2, 1, 3)).reshape(batch * cols, 2 * rows)
    z = oracle.transform_batch(sr, colmajor, direction).reshape(batch, cols, rows, 2)
    oracle.destroy_setup(sr)
    return np.ascontiguousarray(z.transpose(0, 2, 1, 3)).reshape(batch, rows, 2 * cols)


# The second line: enough column tiles that every persistent workgroup of the column pass walks through several (it requests
# its next tile while the current one is in the stages) -- compile-time lists with 256 / 512 / 1024 work-items and 16- / 8-column
# tiles (144, 256, 480 rows; one-image kernels: 640, 1080 rows in 16-column, 1536, 2000, 2048 rows in 8-column tiles), the run-time list (30 rows).
@pytest.mark.parametrize("rows,cols,batch", [(16, 16, 5), (64, 256, 3), (48, 80, 4), (256, 4096, 2), (1024, 64, 2), (720, 48, 2),
                                             (2048, 16, 1), (17 - 1, 32768, 1), (96, 65536, 1), (30, 100 * 16, 3), (9, 4096, 2),
                                             (144, 1024, 64), (256, 256, 256), (480, 640, 16), (1080, 256, 24), (2048, 64, 64), (30, 256, 400),
                                             (1536, 64, 128), (2000, 32, 256), (640, 640, 8)])
def test_2d_vs_oracle_axes_and_float64(gpu, oracle, rows, cols, batch):
    api, torch, dev = gpu
    s = api.fft_new_setup(cols, api.FFT_COMPLEX)
    assert s is not None
    x = ol.uniform(batch * rows * 2 * cols, 3 + rows + cols).reshape(batch, rows, 2 * cols)
    xd = torch.from_numpy(x).to(dev)
    for direction, odir in ((api.FFT_FORWARD, FORWARD), (api.FFT_BACKWARD, BACKWARD)):
        yd = torch.empty_like(xd)
        api.fft_transform_2d_batch(s, rows, xd, yd, batch, direction)
        torch.cuda.synchronize()
        got = yd.cpu().numpy().reshape(batch, rows * 2 * cols)
        z = x[..., 0::2].astype(np.float64) + 1j * x[..., 1::2]
        t = np.fft.fft2(z, axes=(1, 2)) if odir == FORWARD else np.fft.ifft2(z, axes=(1, 2)) * (rows * cols)
        exact = np.empty((batch, rows, 2 * cols))
        exact[..., 0::2], exact[..., 1::2] = t.real, t.imag
        assert_frames_close(got, exact.reshape(batch, -1), what=f"2-D {rows}x{cols} vs float64")
        want = _oracle_2d(oracle, x, rows, cols, odir)
        if want is not None:
            assert_frames_close(got, want.reshape(batch, -1), what=f"2-D {rows}x{cols} vs the oracle along each axis", maxabs_tol=2e-6)
    # in place, and the round trip
    api.fft_transform_2d_batch(s, rows, xd, xd, batch, api.FFT_FORWARD)
    api.fft_transform_2d_batch(s, rows, xd, xd, batch, api.FFT_BACKWARD)
    torch.cuda.synchronize()
    assert float((xd.cpu() / (rows * cols) - torch.from_numpy(x)).abs().max()) < 4e-6
    api.fft_destroy_setup(s)


def test_2d_argument_checks(gpu):
    api, torch, dev = gpu
    s = api.fft_new_setup(64, api.FFT_COMPLEX)
    x = torch.zeros((7, 128), device=dev)
    with pytest.raises(api.FFTError):
        api.fft_transform_2d_batch(s, 7, x, x, 1, api.FFT_FORWARD)  # 7 rows: not 2-3-5-smooth
    with pytest.raises(api.FFTError):
        api.fft_transform_2d_batch(s, 4096, x, x, 1, api.FFT_FORWARD)  # too many rows for a column tile
    r = api.fft_new_setup(64, api.FFT_REAL)
    with pytest.raises(api.FFTError):
        api.fft_transform_2d_batch(r, 4, x, x, 1, api.FFT_FORWARD)
    api.fft_destroy_setup(s)
    api.fft_destroy_setup(r)
