"""``convd``: oracle pins on CPU, device parity on the GPU (reference test: test/KernTest.m:115-163)."""
import numpy as np
import pytest

from oracle import convd_oracle as O


def rel(a, b):
    den = np.abs(b).max() if b.size else 1.0
    return float(np.abs(a - b).max() / (den if den > 0 else 1.0)) if b.size else 0.0


# ---------------------------------------------------------------- CPU: the oracle against the reference's documented answers
def test_oracle_doc_example_autocorrelation():
    a = np.array([1, -2, 3, -4, 5.0])
    z, lags = O.convd(a, a[::-1])                                       # kern/convd.m:33-35
    assert np.array_equal(z, np.correlate(a, a, "full")) and np.array_equal(z, [5, -14, 26, -40, 55, -40, 26, -14, 5])
    assert np.array_equal(lags, np.arange(-4, 5))
    z2, _ = O.convd(a)                                                  # default y = conj(flip(x))
    assert np.array_equal(z2, z)
    c = a + 1j * a[::-1]
    assert np.allclose(O.convd(c)[0], np.correlate(c, c, "full"))       # xcorr of a complex sequence


@pytest.mark.parametrize("M,N", [(4, 2), (5, 3), (7, 4), (3, 5), (6, 1), (1, 1)])
def test_oracle_shapes_follow_matlab_conv(M, N):
    """MATLAB conv(u, v, 'same') = central part, size of u, starting at full index ceil((N-1)/2); 'valid' = full[N-1 : M]"""
    rng = np.random.default_rng(M * 10 + N)
    u, v = rng.standard_normal(M), rng.standard_normal(N)
    f = np.convolve(u, v, "full")
    assert np.allclose(O.convd(u, v, 1, "full")[0], f)
    s0 = -(-(N - 1) // 2)
    assert np.allclose(O.convd(u, v, 1, "same")[0], f[s0:s0 + M])
    assert np.allclose(O.convd(u, v, 1, "valid")[0], f[N - 1:M] if M >= N else [])
    for shp in ("full", "same", "valid"):
        assert np.allclose(O.convd(u, v, 1, shp)[0], O.convd_direct(u, v, shp))
    assert np.array_equal(O.convd([1, 2, 3, 4.0], [1, 1.0], 1, "same")[0], [3, 5, 7, 4])      # MATLAB, not numpy, centring


def test_oracle_exponential_rows_example():
    """kern/convd.m:37-50: row-wise conv(xa(i,:), xb(i,:), 'same') == convd(xa, xb, 2, 'same')"""
    n, m = np.arange(16)[None, :], np.arange(4)[:, None]
    xa, xb = 0.84 ** (n + m), 0.92 ** (n + m)
    z, _ = O.convd(xa, xb, 2, "same")
    for i in range(4):
        f = np.convolve(xa[i], xb[i], "full")
        assert np.allclose(z[i], f[8:24])
    zb, _ = O.convd(np.stack([xa, xa], -1), xb, 2, "same")              # broadcasting over a trailing dimension
    assert np.allclose(zb[..., 0], z) and np.allclose(zb[..., 1], z)


def test_oracle_incompatible_sizes():
    with pytest.raises(ValueError, match="Incompatible sizes"):
        O.convd(np.zeros((4, 3)), np.zeros((2, 2)), 1)


# ---------------------------------------------------------------- GPU parity
def _dev(a, prec, cplx):
    a = np.asarray(a)
    if not cplx:
        a = a.real
    return a.astype({("single", True): np.complex64, ("single", False): np.float32, ("double", True): np.complex128,
                     ("double", False): np.float64}[(prec, cplx)])


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["single", "double"])
@pytest.mark.parametrize("cplx", [True, False])
def test_convd_reference_test_case(prec, cplx):
    """the reference's own case (test/KernTest.m:51-55,152-160): 1024 random samples, every shape, every dimension"""
    import torch
    from qups_amd import convd
    rng = np.random.default_rng(0)
    A = _dev(rng.standard_normal(1024) + 1j * rng.standard_normal(1024), prec, cplx)
    B = _dev(rng.standard_normal(1024) + 1j * rng.standard_normal(1024), prec, cplx)
    tol = 1e4 * np.finfo(np.float32).eps if prec == "single" else 1e4 * np.finfo(np.float64).eps     # KernTest.m:131-135 (x eps(max))
    for shape in ("full", "same", "valid"):
        ref, lags = O.convd(A, B, 1, shape)
        for dim in (1, 2, 3):
            sz = [1] * 3
            sz[dim - 1] = 1024
            z, lg = convd(torch.from_numpy(A.reshape(sz)), torch.from_numpy(B.reshape(sz)), dim, shape, return_lags=True)
            assert z.dtype == torch.from_numpy(A).dtype and tuple(z.shape) == tuple(1 if k != dim - 1 else len(ref) for k in range(3))
            assert rel(z.cpu().numpy().reshape(-1), ref) <= tol, (shape, dim)
            assert np.array_equal(lg.reshape(-1), lags.reshape(-1))
    z = convd(torch.from_numpy(A)).cpu().numpy()                                          # auto-correlation default
    assert rel(z, np.correlate(A.astype(np.complex128), A.astype(np.complex128), "full")) <= tol
    zb = convd(torch.from_numpy(np.stack([A, A], -1).reshape(1024, 1, 1, 2)), torch.from_numpy(B))   # cat(4, A, A) broadcast
    z1 = convd(torch.from_numpy(A), torch.from_numpy(B))
    # (two columns: the direct kernel; the single 1024-tap complex64 trace: the FFT convolution -- other roundings, the same numbers)
    assert torch.equal(zb[:, 0, 0, 0], zb[:, 0, 0, 1]) and rel(zb[:, 0, 0, 0].cpu().numpy(), z1.cpu().numpy()) <= tol


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["full", "same", "valid"])
@pytest.mark.parametrize("sz_x,sz_y,dim", [
    ((1500, 3, 2), (37, 3, 2), 1), ((1500, 3, 2), (37, 1, 1), 1), ((2, 3, 1100), (2, 3, 300), 3), ((2, 3, 1100), (1, 1, 300), 3),
    ((5, 260, 70), (5, 9, 70), 2), ((5, 260, 70), (1, 9, 1), 2), ((5, 260, 70), (5, 9, 1), 2), ((5, 260, 1), (1, 9, 70), 2),
    ((3, 40), (3, 90), 2), ((7, 1), (1, 1), 1), ((2049, 2), (1025, 1), 1), ((4, 3, 5), (4, 1, 5), 2), ((2, 1, 64), (2, 3, 5), 3),
])
def test_convd_shapes_dims_and_broadcast(sz_x, sz_y, dim, shape):
    import torch
    from qups_amd import convd
    rng = np.random.default_rng(hash((sz_x, sz_y, dim)) & 0xffff)
    x = (rng.standard_normal(sz_x) + 1j * rng.standard_normal(sz_x)).astype(np.complex64)
    y = (rng.standard_normal(sz_y) + 1j * rng.standard_normal(sz_y)).astype(np.complex64)
    ref, lags = O.convd(x, y, dim, shape)
    z, lg = convd(torch.from_numpy(x), torch.from_numpy(y), dim, shape, return_lags=True)
    assert tuple(z.shape) == ref.shape and lg.shape == lags.shape and np.array_equal(lg, lags)
    assert rel(z.cpu().numpy(), ref) <= 2e-5
    xr, yr = x.real.copy(), y.real.copy()                                                  # real data, mixed precision -> single
    zr = convd(torch.from_numpy(xr.astype(np.float64)), torch.from_numpy(yr), dim, shape)
    assert zr.dtype == torch.float32 and rel(zr.cpu().numpy(), O.convd(xr, yr, dim, shape)[0]) <= 2e-5
    zm = convd(torch.from_numpy(x), torch.from_numpy(yr), dim, shape)                        # complex traces, real taps: the taps stay real on the device
    assert zm.dtype == torch.complex64 and rel(zm.cpu().numpy(), O.convd(x, yr, dim, shape)[0]) <= 2e-5
    zh = convd(torch.from_numpy(x), torch.from_numpy(yr).to(torch.float16), dim, shape)     # ... also in half precision (convch)
    assert zh.dtype == torch.complex32
    zh = torch.view_as_complex(torch.view_as_real(zh).float()).cpu().numpy()
    x16 = x.real.astype(np.float16).astype(np.float64) + 1j * x.imag.astype(np.float16).astype(np.float64)
    assert rel(zh, O.convd(x16, yr.astype(np.float16).astype(np.float64), dim, shape)[0]) <= 2e-3


@pytest.mark.gpu
def test_convd_band_pass_of_channel_data():
    """the use in front of DAS: FIR band-pass of every trace (T x N x M, time first) == per-trace numpy convolution"""
    import torch
    from qups_amd import convd
    rng = np.random.default_rng(3)
    T, N, M, K = 700, 8, 5, 63
    x = rng.standard_normal((T, N, M)).astype(np.float32)
    n = np.arange(K) - (K - 1) / 2
    h = (np.sinc(0.5 * n) - 0.5 * np.sinc(0.25 * n) * 0.5) * np.hamming(K)
    z = convd(torch.from_numpy(x).cuda(), torch.from_numpy(h.astype(np.float32)).cuda(), 1, "same").cpu().numpy()
    ref = np.stack([[np.convolve(x[:, a, b].astype(np.float64), h, "full")[(K - 1) - (K - 1) // 2:][:T] for b in range(M)] for a in range(N)], 0).transpose(2, 0, 1)
    assert z.shape == x.shape and rel(z, ref) <= 1e-5
    zt = convd(torch.from_numpy(np.ascontiguousarray(x.transpose(2, 1, 0))).cuda(), torch.from_numpy(h.astype(np.float32)).cuda().reshape(1, 1, K), 3, "same")
    assert rel(zt.cpu().numpy().transpose(2, 1, 0), ref) <= 1e-5                         # time-contiguous layout: the LDS kernel


@pytest.mark.gpu
def test_convd_errors_and_empty():
    import torch
    from qups_amd import convd
    with pytest.raises(ValueError, match="Incompatible sizes"):
        convd(torch.zeros(4, 3), torch.zeros(2, 2), 1)
    with pytest.raises(ValueError, match="shape must be one of"):
        convd(torch.zeros(4), torch.zeros(2), 1, "circular")
    z = convd(torch.ones(3), torch.ones(5), 1, "valid")
    assert tuple(z.shape) == (0,)
    with pytest.raises(TypeError):
        convd(torch.ones(3, dtype=torch.int32), torch.ones(2))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("QDAS_CONVD_FUZZ", "48"))))
def test_convd_random_configuration(seed):
    """random ranks, dimensions, broadcast patterns, lengths (around the 1024-output tile / 256-tap chunk), types"""
    import torch
    from qups_amd import convd
    r = np.random.default_rng(7000 + seed)
    D = int(r.integers(1, 5))
    d = int(r.integers(0, D))
    full = [int(r.choice([1, 2, 3, 5, 70])) for _ in range(D)]
    M = int(r.choice([1, 2, 7, 255, 256, 257, 1023, 1024, 1025, 1300, 2400]))
    N = int(r.choice([1, 2, 3, 4, 5, 31, 64, 255, 256, 257, 300, 700]))
    sx, sy = list(full), list(full)
    sx[d], sy[d] = M, N
    for k in range(D):
        if k != d and r.integers(0, 3) == 0:
            (sx if r.integers(0, 2) else sy)[k] = 1
    while int(np.prod(sx)) * N > 3e8 or int(np.prod(sy)) * M > 3e8:       # keep the oracle fast
        k = int(np.argmax([v if i != d else 0 for i, v in enumerate(full)]))
        full[k] = sx[k] = sy[k] = 1
    shape = str(r.choice(["full", "same", "valid"]))
    cx, cy, dbl = bool(r.integers(0, 2)), bool(r.integers(0, 2)), bool(r.integers(0, 4) == 0)
    rt = np.float64 if dbl else np.float32
    mk = lambda sz, cp: ((r.standard_normal(sz) + 1j * r.standard_normal(sz)).astype(np.complex128 if dbl else np.complex64) if cp
                         else r.standard_normal(sz).astype(rt))
    x, y = mk(sx, cx), mk(sy, cy)
    if M * max(1, int(np.prod(full)) // max(full[d], 1)) > 2e6 and N > 300:
        pytest.skip("oracle too slow")
    ref, lags = O.convd(x, y, d + 1, shape)
    z, lg = convd(torch.from_numpy(x), torch.from_numpy(y), d + 1, shape, return_lags=True)
    assert tuple(z.shape) == ref.shape, (sx, sy, d, shape)
    assert z.is_complex() == (cx or cy) and (z.dtype in (torch.float64, torch.complex128)) == dbl
    assert np.array_equal(lg, lags)
    assert rel(z.cpu().numpy(), ref) <= (1e-12 if dbl else 3e-5), (seed, sx, sy, d, shape, cx, cy, dbl)


@pytest.mark.gpu
@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("sz_x,sz_y,dim", [((1500, 3, 2), (37, 1, 1), 1), ((2, 3, 1100), (1, 1, 300), 3), ((5, 260, 70), (5, 9, 1), 2), ((2049,), (65,), 1)])
def test_convd_half_precision(sz_x, sz_y, dim, cplx):
    """the half-precision twins convh / convch (reference src/convd.cu:141,153; dispatch kern/convd.m:263-265): half in, half out, products
    and sums in fp32 -- against the float64 oracle on the half-rounded inputs, to an output ulp"""
    import torch
    from qups_amd import convd
    rng = np.random.default_rng(11)
    h16 = lambda a: a.astype(np.float16)
    xr, xi, yr, yi = (h16(rng.standard_normal(s) * 0.25) for s in (sz_x, sz_x, sz_y, sz_y))
    if cplx:
        x64, y64 = xr.astype(np.float64) + 1j * xi, yr.astype(np.float64) + 1j * yi
        to_t = lambda re, im: torch.view_as_complex(torch.from_numpy(np.stack([re, im], -1)).contiguous())
        xt, yt = to_t(xr, xi), to_t(yr, yi)
    else:
        x64, y64 = xr.astype(np.float64), yr.astype(np.float64)
        xt, yt = torch.from_numpy(xr), torch.from_numpy(yr)
    for shape in ("full", "same", "valid"):
        ref, lags = O.convd(x64, y64, dim, shape)
        z, lg = convd(xt, yt, dim, shape, return_lags=True)
        assert z.dtype == (torch.complex32 if cplx else torch.float16) and np.array_equal(lg, lags)
        zn = torch.view_as_real(z).float().cpu().numpy().view(np.complex64)[..., 0] if cplx else z.float().cpu().numpy()
        assert zn.shape == ref.shape
        assert np.abs(zn - ref).max() <= 1.5e-3 * np.abs(ref).max(), shape


@pytest.mark.gpu
def test_channeldata_filter_and_downsample():
    """ChannelData.filter with an FIR filter (reference src/ChannelData.m:857-888: filter(D, x) along time, t0 -= (order / 2) / fs) and
    ChannelData.downsample (:1042-1058): against scipy.signal.lfilter per trace; every time-dimension position, real and complex data"""
    import torch
    from scipy.signal import firwin, lfilter
    from qups_amd.ultrasound import ChannelData
    rng = np.random.default_rng(5)
    T, N, M, fs = 900, 6, 4, 20e6
    x = (rng.standard_normal((T, N, M)) + 1j * rng.standard_normal((T, N, M))).astype(np.complex64)
    b = firwin(26, 0.3)                                        # designfilt('lowpassfir', 'FilterOrder', 25, ...): 26 coefficients
    chd = ChannelData(torch.from_numpy(x).cuda(), -1e-6, fs)
    out = chd.filter(b)
    ref = lfilter(b, 1.0, x.astype(np.complex128), axis=0)
    assert tuple(out.data.shape) == x.shape and rel(out.data.cpu().numpy(), ref) <= 2e-5
    assert abs(out.t0 - (-1e-6 - 12.5 / fs)) < 1e-15 and out.fs == fs
    t0v = np.linspace(0, 1e-6, M).reshape(1, 1, M)             # per-transmit start times move along
    out2 = ChannelData(torch.from_numpy(x.real.copy()).cuda(), t0v, fs).filter(b)
    assert rel(out2.data.cpu().numpy(), ref.real) <= 2e-5 and np.allclose(out2.t0, t0v - 12.5 / fs)
    perm = ChannelData(torch.from_numpy(np.ascontiguousarray(x.transpose(1, 2, 0))).cuda(), 0.0, fs, "NMT").filter(b)       # time last: the LDS kernel
    assert rel(perm.data.cpu().numpy().transpose(2, 0, 1), ref) <= 2e-5
    ds = out.downsample(4)
    assert ds.fs == fs / 4 and ds.t0 == out.t0 and torch.equal(ds.data, out.data[::4]) and ds.data.is_contiguous()
    with pytest.raises(ValueError):
        out.downsample(0)


@pytest.mark.gpu
@pytest.mark.parametrize("taps_complex", [False, True], ids=["real-taps", "complex-taps"])
@pytest.mark.parametrize("T,K", [(700, 129), (2816, 129), (1000, 97), (2048, 301), (5000, 500), (90, 200)])
def test_convd_long_filters_take_the_fft_path(T, K, taps_complex, monkeypatch):
    """complex64 traces, time contiguous, one filter of >= 128 taps (QDAS_CONV_FFT_MIN_TAPS): the FFT convolution with the trace resident in LDS (csrc/pre.hip fftconv_launch)
    -- every shape against numpy's direct sum in float64 and against the direct kernel (QDAS_CONV_FFT_MIN_TAPS switches the path per call)"""
    import torch
    from qups_amd import convd
    rng = np.random.default_rng(T + K)
    S = 37
    x = (rng.standard_normal((S, T)) + 1j * rng.standard_normal((S, T))).astype(np.complex64)
    h = rng.standard_normal(K) * np.hanning(K)
    h = (h + 1j * rng.standard_normal(K) * np.hanning(K)).astype(np.complex64) if taps_complex else h.astype(np.float32)
    xt, ht = torch.from_numpy(x).cuda(), torch.from_numpy(h).cuda().reshape(1, K)
    full = np.stack([np.convolve(x[s].astype(np.complex128), h.astype(np.complex128 if taps_complex else np.float64), "full") for s in range(S)])
    for shape in ("full", "same", "valid"):
        if shape == "valid" and T < K:
            continue
        off = {"full": 0, "same": (K - 1) - (K - 1) // 2, "valid": K - 1}[shape]
        L = {"full": T + K - 1, "same": T, "valid": T - K + 1}[shape]
        ref = full[:, off:off + L]
        monkeypatch.setenv("QDAS_CONV_FFT_MIN_TAPS", "90")
        z = convd(xt, ht, 2, shape).cpu().numpy()
        monkeypatch.setenv("QDAS_CONV_FFT_MIN_TAPS", "1000000")
        zd = convd(xt, ht, 2, shape).cpu().numpy()
        assert z.shape == ref.shape and rel(z, ref) <= 5e-6, (shape, rel(z, ref))     # (fp32 transforms of up to 8192 points: a few 1e-6 of the largest output)
        assert rel(zd, ref) <= 5e-6 and rel(z, zd) <= 6e-6
        if T + K - 1 <= 8192:
            assert not np.array_equal(z, zd), "the two paths round differently: identical bits mean the FFT path did not run"


# ---------------------------------------------------------------- IIR filtering (ChannelData.filter with an IIR digitalFilter: qdas_iir)
def test_oracle_sosfilt_is_matlab_filter():
    """the restated cascade of direct-form II transposed sections against scipy.signal.sosfilt and -- as a transfer function -- lfilter (= MATLAB filter(b, a, x))"""
    from scipy import signal
    rng = np.random.default_rng(4)
    x = rng.standard_normal((300, 5)) + 1j * rng.standard_normal((300, 5))
    for order, wn, kind in ((4, 0.2, "low"), (6, [0.1, 0.4], "band"), (3, 0.3, "high")):
        sos = signal.butter(order, wn, kind, output="sos")
        assert rel(O.sosfilt(x, sos), signal.sosfilt(sos, x, axis=0)) <= 1e-12
        b, a = signal.butter(order, wn, kind)
        assert rel(O.sosfilt(x, sos), signal.lfilter(b, a, x, axis=0)) <= 1e-9
    sos1 = np.array([[0.5, 0.25, 0.0, 2.0, -0.6, 0.0]])                 # a first-order section with a0 != 1
    assert rel(O.sosfilt(x.real, sos1, gain=3.0), 3.0 * signal.lfilter([0.25, 0.125], [1.0, -0.3], x.real, axis=0)) <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,cplx", [("float32", True), ("float32", False), ("float64", True), ("float64", False)])
def test_sosfilt_on_the_device(dtype, cplx):
    """qdas_iir against the oracle: ragged trace counts (not a multiple of 64), record lengths that are not a multiple of the 32-sample tile, along either dimension,
    8th-order band-pass and a first-order section with a gain; fp32 data with double state: 1e-6 of the peak"""
    import torch
    from scipy import signal
    from qups_amd import sosfilt
    rng = np.random.default_rng(7)
    for shape, dim, sos, gain in (((301, 70), 1, signal.butter(4, [0.1, 0.4], "band", output="sos"), 1.0),
                                  ((3, 129, 5), 2, signal.cheby1(5, 1.0, 0.3, output="sos"), 0.5),
                                  ((33, 1), 1, np.array([[0.5, 0.25, 0.0, 2.0, -0.6, 0.0]]), 3.0)):
        x = rng.standard_normal(shape)
        if cplx:
            x = x + 1j * rng.standard_normal(shape)
        x = x.astype({"float32": np.complex64 if cplx else np.float32, "float64": np.complex128 if cplx else np.float64}[dtype])
        y = sosfilt(torch.from_numpy(x), sos, dim, gain).cpu().numpy()
        ref = O.sosfilt(x.astype(np.complex128 if cplx else np.float64), sos, dim - 1, gain)
        assert y.shape == x.shape and rel(y, ref) <= (1e-6 if dtype == "float32" else 1e-13), (shape, dim, rel(y, ref))
    # an EXPLICIT zero gain is zero output (the C descriptor reads 0 as "not set" = 1: ADVICE r5)
    z = sosfilt(torch.from_numpy(x), sos, dim, 0.0).cpu().numpy()
    assert z.shape == x.shape and not z.any()


@pytest.mark.gpu
def test_channeldata_filter_with_an_iir_filter():
    """ChannelData.filter with second-order sections / a transfer function: the data are filtered along time and t0 moves by filtord / fs (reference src/ChannelData.m:876-884)"""
    import torch
    from scipy import signal
    from qups_amd import ChannelData
    rng = np.random.default_rng(9)
    x = (rng.standard_normal((200, 6, 4)) + 1j * rng.standard_normal((200, 6, 4))).astype(np.complex64)
    chd = ChannelData(torch.from_numpy(x), 1e-6, 20e6)
    sos = signal.butter(3, 0.25, output="sos")                          # order 3: one second-order + one first-order section
    out = chd.filter(None, sos=sos)
    assert abs(out.t0 - (1e-6 - 3 / 20e6)) < 1e-15
    assert rel(out.data.cpu().numpy(), O.sosfilt(x.astype(np.complex128), sos, 0)) <= 1e-6
    b, a = signal.butter(3, 0.25)
    out2 = chd.filter(b, a=a)
    assert abs(out2.t0 - out.t0) < 1e-15 and rel(out2.data.cpu().numpy(), signal.lfilter(b, a, x.astype(np.complex128), axis=0)) <= 1e-6
