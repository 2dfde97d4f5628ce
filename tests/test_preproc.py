"""hilbert (+ downmix) in front of the DAS path: oracle = numpy FFT restatement of ChannelData.hilbert
(reference src/ChannelData.m:960-964) and ChannelData.downmix (:765)."""
from __future__ import annotations

import numpy as np
import pytest


def hilbert_ref(x, N=None, fdown=0.0, t0=0.0, fs=1.0):
    x = np.asarray(x, np.float64)
    T = x.shape[0]
    N = T if N is None else N
    X = np.fft.fft(x, N, axis=0)                                       # :960
    Nd2 = N // 2                                                       # :961
    w = np.concatenate([[1.0], 2.0 * np.ones(Nd2 - 1), [1.0 + N % 2], np.zeros(N - Nd2 - 1)])[:N]   # :962
    y = np.fft.ifft(X * w.reshape((N,) + (1,) * (x.ndim - 1)), axis=0)  # :963-964
    if fdown:
        t = (t0 + np.arange(N) / fs).reshape((N,) + (1,) * (x.ndim - 1))
        y = y * np.exp(-2j * np.pi * fdown * t)                        # :765
    return y


def test_reference_weights_equal_the_textbook_analytic_signal():
    rng = np.random.default_rng(0)
    for T in (16, 17, 100, 255):
        x = rng.standard_normal((T, 3))
        y = hilbert_ref(x)
        assert np.abs(y.real - x).max() <= 1e-12                      # the analytic signal keeps the input as its real part
        X = np.fft.fft(y, axis=0)
        assert np.abs(X[T // 2 + 1:]).max() <= 1e-10 * np.abs(X).max()  # ... and has no negative frequencies


@pytest.mark.gpu
@pytest.mark.parametrize("T,N,dtype,fdown", [(256, None, "f32", 0.0), (300, 512, "f32", 0.0), (301, None, "f32", 0.0), (400, 256, "i16", 0.0),
                                             (384, 512, "f32", 4.0e6), (1000, 1024, "i16", 5.0e6)])
def test_hilbert_matches_numpy(T, N, dtype, fdown):
    from qups_amd.preproc import hilbert
    rng = np.random.default_rng(1)
    x = rng.standard_normal((T, 5, 3))
    fs, t0 = 20e6, -3.2e-6
    if dtype == "i16":
        xq = np.round(x * 2000).astype(np.int16)
        ref = hilbert_ref(xq.astype(np.float64), N, fdown, t0, fs)
    else:
        xq = x.astype(np.float32)
        ref = hilbert_ref(xq.astype(np.float64), N, fdown, t0, fs)
    y = hilbert(xq, N, fdown, t0, fs).cpu().numpy()
    assert y.shape == ref.shape and y.dtype == np.complex64
    assert np.abs(y - ref).max() / np.abs(ref).max() <= 2e-5


@pytest.mark.gpu
def test_real_rf_hilbert_then_das_equals_das_of_the_analytic_data():
    """the reference's pipeline (example_.m:261-269): real traces -> hilbert -> DAS, all on the device"""
    import torch
    from qups_amd import das_spec
    from qups_amd.preproc import hilbert
    from tests.cases import make_case, rel_err
    case = make_case(seq="PW", interp="cubic", seed=95, N=16, I1=120, I2=16, zlim=(4e-3, 14e-3), xspan=3e-3)
    rf = np.ascontiguousarray(case["x"].real.astype(np.float32))              # what a scanner delivers
    xa = hilbert(rf)                                                     # T x N x M complex64, on the device
    ref_x = hilbert_ref(rf.astype(np.float64)).astype(np.complex64)
    args = (case["Pi"], case["Pr"], case["Pv"], case["Nv"])
    opts = list(case["opt"]) + ["interp", "cubic"]
    b_dev = das_spec("DAS", *args, xa, case["t0"], case["fs"], case["c"], *opts)
    b_ref = das_spec("DAS", *args, torch.from_numpy(ref_x), case["t0"], case["fs"], case["c"], *opts)
    torch.cuda.synchronize()
    assert rel_err(b_dev.cpu().numpy(), b_ref.cpu().numpy()) <= 2e-5
