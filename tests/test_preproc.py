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
@pytest.mark.parametrize("force_hipfft", [False, True])
@pytest.mark.parametrize("T,N,K,dtype,fdown", [(2816, None, 7, "f32", 0.0),       # C3's record length = 2^8 * 11
                                               (2048, None, 6, "i16", 5.0e6),     # C1 / C2
                                               (1000, None, 3, "f32", 0.0),       # 2^3 5^3
                                               (143, None, 5, "f32", 0.0),        # 11 * 13, odd length, odd trace count
                                               (2, None, 4, "f32", 0.0),
                                               (252, 5632, 3, "f32", 2.0e6),      # 2^9 11: 11 8 8 8, 704 threads
                                               (1000, 1820, 3, "f32", 0.0),       # 2^2 5 7 13
                                               (4000, 3645, 2, "f32", 0.0),       # truncation; 3^6 * 5
                                               (300, 8192, 1, "i16", 0.0),        # 16 8 8 8, 1024 threads
                                               (9, None, 2, "f32", 0.0), (16, None, 3, "f32", 0.0), (512, None, 2, "f32", 0.0)])
def test_one_pass_hilbert_matches_numpy(T, N, K, dtype, fdown, force_hipfft, monkeypatch):
    """the LDS-resident kernel (two real traces per complex transform, mixed-radix stages LDS to LDS) against the numpy restatement,
    and the hipFFT passes it replaces against the same numbers"""
    from qups_amd.preproc import hilbert
    monkeypatch.setenv("QDAS_PRE_HIPFFT", "1" if force_hipfft else "0")
    rng = np.random.default_rng(T + K)
    x = rng.standard_normal((T, K))
    fs, t0 = 20e6, 1.7e-6
    xq = np.round(x * 3000).astype(np.int16) if dtype == "i16" else x.astype(np.float32)
    ref = hilbert_ref(xq.astype(np.float64), N, fdown, t0, fs)
    y = hilbert(xq, N, fdown, t0, fs).cpu().numpy()
    assert hilbert.last_one_pass == (not force_hipfft)
    assert y.shape == ref.shape and y.dtype == np.complex64
    assert np.abs(y - ref).max() / np.abs(ref).max() <= 2e-5
    if not fdown:
        Nn = ref.shape[0]
        pad = np.zeros((Nn, K), np.float32); pad[:min(T, Nn)] = xq[:min(T, Nn)]
        assert np.array_equal(y.real, pad)                              # the real part IS the input, bit for bit


@pytest.mark.gpu
def test_hilbert_lengths_the_one_pass_kernel_does_not_serve_fall_to_hipfft():
    from qups_amd.preproc import hilbert
    rng = np.random.default_rng(5)
    for T in (301, 2 * 17, 8190, 10240):                               # 7 * 43; a factor 17; 2 3^2 5 7 13 (4095 radix-2 butterflies); too long
        x = rng.standard_normal((T, 3)).astype(np.float32)
        y = hilbert(x).cpu().numpy()
        assert not hilbert.last_one_pass
        ref = hilbert_ref(x.astype(np.float64))
        assert np.abs(y - ref).max() / np.abs(ref).max() <= 2e-5


@pytest.mark.gpu
def test_hilbert_takes_time_fastest_views_without_a_copy():
    import torch
    from qups_amd.preproc import hilbert
    g = torch.Generator(device="cuda").manual_seed(3)
    xkt = torch.randn((6, 5, 704), generator=g, device="cuda")          # memory: M x N x T  ==  MATLAB's T x N x M
    x = xkt.permute(2, 1, 0)                                            # T x N x M view, time fastest
    y = hilbert(x)
    ref = hilbert_ref(x.cpu().numpy().astype(np.float64))
    assert y.shape == (704, 5, 6)
    assert np.abs(y.cpu().numpy() - ref).max() / np.abs(ref).max() <= 2e-5


@pytest.mark.gpu
def test_hilbert_of_a_strided_single_trace():
    """hilbert(x[:, j]) of a T x N tensor and a decimated slice x[::2]: one trace (K == 1) whose samples are NOT consecutive in memory
    (the K == 1 fast path handed the raw pointer over and the library read the wrong samples)"""
    import torch
    from qups_amd.preproc import hilbert
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn((512, 3), generator=g, device="cuda")
    for v in (x[:, 1], x[::2, 2], x[:, 0].contiguous(), x.t().contiguous()[1]):
        y = hilbert(v)
        ref = hilbert_ref(v.cpu().numpy().astype(np.float64))
        assert y.shape == v.shape
        assert np.abs(y.cpu().numpy() - ref).max() / np.abs(ref).max() <= 2e-5, v.stride()


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


@pytest.mark.gpu
def test_channeldata_hilbert_downmix_downsample_chain():
    """the reference's demodulation chain (src/ChannelData.m:757-800 example: hilbert -> downmix -> downsample) through the ChannelData mirror:
    against the numpy restatement, for time first and time last, scalar and per-transmit start times"""
    import torch
    from qups_amd.ultrasound import ChannelData
    rng = np.random.default_rng(8)
    T, N, M, fs, fc = 1024, 6, 5, 20e6, 5e6
    x = rng.standard_normal((T, N, M)).astype(np.float32)
    t0 = -1.3e-6
    chd = ChannelData(torch.from_numpy(x).cuda(), t0, fs)
    a = chd.hilbert()
    ref = hilbert_ref(x.astype(np.float64))
    assert a.data.dtype == torch.complex64 and np.abs(a.data.cpu().numpy() - ref).max() / np.abs(ref).max() <= 2e-5
    b = a.downmix(fc)
    tt = (t0 + np.arange(T) / fs).reshape(T, 1, 1)
    refb = ref * np.exp(-2j * np.pi * fc * tt)
    assert np.abs(b.data.cpu().numpy() - refb).max() / np.abs(refb).max() <= 2e-5
    c = b.downsample(4)
    assert c.fs == fs / 4 and np.abs(c.data.cpu().numpy() - refb[::4]).max() / np.abs(refb).max() <= 2e-5
    t0v = np.linspace(-1e-6, 1e-6, M).reshape(1, 1, M)                                    # one start time per transmit
    bv = ChannelData(a.data, t0v, fs).downmix(fc)
    refv = ref * np.exp(-2j * np.pi * fc * (t0v + np.arange(T).reshape(T, 1, 1) / fs))
    assert np.abs(bv.data.cpu().numpy() - refv).max() / np.abs(refv).max() <= 2e-5
    p = ChannelData(torch.from_numpy(np.ascontiguousarray(x.transpose(1, 2, 0))).cuda(), t0, fs, "NMT").hilbert(1280)      # time last, zero-padded
    refp = hilbert_ref(x.astype(np.float64), 1280)
    assert tuple(p.data.shape) == (N, M, 1280) and np.abs(p.data.cpu().numpy().transpose(2, 0, 1) - refp).max() / np.abs(refp).max() <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("QDAS_PRE_FUZZ", "32"))))
def test_hilbert_random_lengths_and_batches(seed):
    """random record lengths (smooth numbers for the one-pass kernel -- fixed stage lists, run-time lists, the 512-thread variant -- and arbitrary
    ones for the hipFFT passes), transform lengths, trace counts, input types and downmix frequencies"""
    from qups_amd.preproc import hilbert
    r = np.random.default_rng(9000 + seed)
    if r.integers(0, 3):
        N = 1
        while N < 2 or N > 8192:                                       # a random smooth number
            N = int(np.prod(r.choice([2, 2, 2, 2, 3, 3, 5, 7, 11, 13], size=int(r.integers(1, 11)))))
    else:
        N = int(r.integers(2, 9000))
    T = int(max(1, N + r.integers(-N // 2, N // 2 + 1))) if r.integers(0, 2) else N
    K = int(r.choice([1, 2, 3, 8, 33]))
    i16 = bool(r.integers(0, 2))
    fdown = float(r.choice([0.0, 0.0, 3.3e6, -1.7e6]))
    x = r.standard_normal((T, K))
    xq = np.round(x * 1500).astype(np.int16) if i16 else x.astype(np.float32)
    fs, t0 = 25e6, -2.1e-6
    y = hilbert(xq, N if N != T else None, fdown, t0, fs).cpu().numpy()
    ref = hilbert_ref(xq.astype(np.float64), N, fdown, t0, fs)
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() / max(np.abs(ref).max(), 1e-30) <= 3e-5, (N, T, K, i16, fdown, hilbert.last_one_pass)


@pytest.mark.gpu
def test_hilbert_plans_are_cached_by_shape(monkeypatch):
    """a frame loop calls ``hilbert`` with ONE record shape -- the plan (for lengths outside the one-pass kernel: two hipFFT plans and three work
    buffers, 11-15 ms to create) is made once and reused; another shape, another plan; the cache is bounded; results do not depend on it."""
    import torch
    from qups_amd import preproc
    from qups_amd.preproc import hilbert
    monkeypatch.setenv("QDAS_PRE_HIPFFT", "1")                        # (the hipFFT passes: where plan creation dominated)
    preproc.clear_pre_plan_cache()
    hilbert.plans_created = 0
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.standard_normal((1031, 24)).astype(np.float32)).cuda()      # 1031 is prime: never the one-pass kernel
    y0 = hilbert(x)
    for _ in range(5):
        y = hilbert(x)
    assert hilbert.plans_created == 1 and not hilbert.last_one_pass and torch.equal(y, y0)
    hilbert(x[:, :7].contiguous())
    assert hilbert.plans_created == 2
    for k in range(preproc._PLAN_CACHE_MAX + 2):                       # more shapes than the cache holds: the oldest are destroyed, the newest stay
        hilbert(x[: 500 + k].contiguous())
    assert len(preproc._PLANS) == preproc._PLAN_CACHE_MAX
    ref = np.asarray(__import__("scipy.signal", fromlist=["hilbert"]).hilbert(x.cpu().numpy().astype(np.float64), axis=0))
    assert np.abs(y0.cpu().numpy() - ref).max() / np.abs(ref).max() <= 3e-5
    preproc.clear_pre_plan_cache()
    assert not preproc._PLANS
