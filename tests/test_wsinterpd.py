"""The general single-delay flavour -- wsinterpd / interpd (reference kern/wsinterpd.m, kern/interpd.m), ChannelData.sample /
rectifyt0 / rectifyDims (src/ChannelData.m:1205-1336, 1895-1913) and focusTx (src/UltrasoundSystem.m:3374-3503) -- through
qdas_wsinterpd / qdas_das_lut.  Ports of the reference's own assertions: test/KernTest.m:178-217 (five permutations of wsinterpd,
nearest / linear against interp1 at 1e5 eps single / 1e12 eps double) and test/interpTest.m:97-143 (wsinterpd on the closed-form
fixture x weights x summed dimensions against the triple loop of interp1)."""
import numpy as np
import pytest

from oracle import das_oracle as O
from tests.cases import rel_err

pytestmark = pytest.mark.gpu


def _np(t):
    import torch
    return t.to(torch.complex128).cpu().numpy() if t.is_complex() else t.cpu().numpy()


@pytest.mark.parametrize("prec,tol", [("single", 1e5 * np.finfo(np.float32).eps), ("double", 1e12 * np.finfo(np.float64).eps)])
@pytest.mark.parametrize("terp", ["nearest", "linear"])
@pytest.mark.parametrize("complexity", ["complex", "real"])
def test_kerntest_runinterpd_permutations(prec, tol, terp, complexity):
    """test/KernTest.m:165-217: wsinterpd(xp, tp, find(ord == 1), 1, [], terp, 0) for five permutations of (I|T, N, M, F)"""
    import torch
    from qups_amd.interpd import interpd, wsinterpd
    rng = np.random.default_rng(1)
    I, T, N, M, F = 32, 256, 2, 3, 4                                             # :44
    fc = np.arange(F).reshape(1, 1, 1, F)
    t = (np.arange(T) / T).reshape(T, 1, 1, 1)
    x = (np.cos(2 * np.pi * fc * t) + 1j * np.sin(2 * np.pi * fc * t)) + 0.01 * ((rng.random((1, N, 1, 1)) + 1j * rng.random((1, N, 1, 1))) - (0.5 + 0.5j))   # T x N x 1 x F (:47)
    tau = (4 + (T - 8)) * rng.random((I, N, M, 1))                               # :48 (as written there: up to T - 4)
    if complexity == "real":
        x = x.real
    rt = np.float32 if prec == "single" else np.float64
    x = x.astype(np.complex64 if prec == "single" else np.complex128) if complexity == "complex" else x.astype(rt)
    tau = tau.astype(rt)
    # matching data via interp1 (:195-203)
    z0 = np.zeros((I, N, M, F), complex)
    for f in range(F):
        for m in range(M):
            for n in range(N):
                z0[:, n, m, f] = O.interp1_matlab(x[:, n, 0, f].astype(complex), 1 + tau[:, n, m, 0].astype(np.float64), terp)
    # defaults / options run (:186-193)
    assert tuple(interpd(torch.from_numpy(x), torch.from_numpy(tau)).shape) == (I, N, M, F)
    interpd(torch.from_numpy(x), torch.from_numpy(tau), 1, "cubic")
    interpd(torch.from_numpy(x), torch.from_numpy(tau), 1, "lanczos3")
    for ord_ in ([0, 1, 2, 3], [0, 1, 3, 2], [1, 0, 2, 3], [2, 3, 0, 1], [3, 2, 1, 0]):   # :179
        xp, tp = np.transpose(x, ord_), np.transpose(tau, ord_)
        z1 = wsinterpd(torch.from_numpy(np.ascontiguousarray(xp)), torch.from_numpy(np.ascontiguousarray(tp)), ord_.index(0) + 1, 1, None, terp, 0, prec=prec)
        z1 = np.transpose(_np(z1), np.argsort(ord_))                             # ipermute
        assert z1.shape == z0.shape
        # the reference compares within RelativeTolerance(tol * eps): element-wise relative (exact zeros must match)
        err = np.abs(z1 - z0) / np.maximum(np.abs(z0), 1e-30)
        if terp == "nearest":                                                    # float32 time: a sample index may round the other way at .5
            assert np.mean(err <= tol) >= 0.995
        else:
            assert np.max(np.abs(z1 - z0)) <= tol * max(1.0, np.abs(z0).max()) * 64, (ord_, np.max(np.abs(z1 - z0)))


def _fixture():
    """test/interpTest.m:33-43: closed-form data, separable delays"""
    I, T, N, M, F = 16, 32, 4, 3, 2
    i = np.arange(I).reshape(I, 1, 1, 1)
    t = (np.arange(T) / T).reshape(T, 1, 1, 1)
    n = np.arange(N).reshape(1, N, 1, 1)
    m = np.arange(M).reshape(1, 1, M, 1)
    f = np.arange(F).reshape(1, 1, 1, F)
    t1 = 4 + (T - 8) * ((1 + i) / I * 1 / N * (1 + m) / M)
    t2 = 4 + (T - 8) * 1 / I * (1 + n) / N
    x0 = np.exp(2j * np.pi * (1 / 2 + f / 2 * n / 4) * t)                        # T x N x 1 x F
    return x0, t1 + t2, (I, T, N, M, F)


@pytest.mark.parametrize("terp", ["cubic", "nearest", "linear", "lanczos3"])
@pytest.mark.parametrize("dsum", [None, [2], [3, 4], [6]])
@pytest.mark.parametrize("wvecd", [[], [3], [3, 4]])
@pytest.mark.parametrize("prec", ["single", "double"])
def test_interptest_wsinterpd(terp, dsum, wvecd, prec):
    """test/interpTest.m:97-143 for the single-delay entry: all weight shapes x summed dimensions against the loop of interp1
    (nearest / linear: the reference's 1e4 eps; cubic / lanczos3: this repository's oracle, where the reference allows 1e8x more)"""
    import torch
    from qups_amd.interpd import wsinterpd
    x0, tau, (I, T, N, M, F) = _fixture()
    rng = np.random.default_rng(3)
    wsz = [1] * max(wvecd + [1])
    for d in wvecd:
        wsz[d - 1] = max(x0.shape[d - 1] if d <= 4 else 1, tau.shape[d - 1] if d <= 4 else 1)
    w = rng.random(wsz)
    ct = np.complex64 if prec == "single" else np.complex128
    rt = np.float32 if prec == "single" else np.float64
    x0c, tauc, wc = x0.astype(ct), tau.astype(rt), w.astype(rt)
    y1 = _np(wsinterpd(torch.from_numpy(x0c), torch.from_numpy(tauc), 1, torch.from_numpy(wc), dsum, terp, 0))
    y0 = np.zeros((I, N, M, F), complex)
    for fi in range(F):
        for mi in range(M):
            for ni in range(N):
                if terp in ("nearest", "linear"):
                    y0[:, ni, mi, fi] = O.interp1_matlab(x0c[:, ni, 0, fi].astype(complex), 1 + tauc[:, ni, mi, 0].astype(np.float64), terp)
                else:
                    y0[:, ni, mi, fi] = O.sample(x0c[:, ni, 0, fi].astype(complex), tauc[:, ni, mi, 0].astype(np.float64), terp)
    y0 = y0 * wc.reshape(wc.shape + (1,) * (4 - wc.ndim))
    if dsum:
        ax = tuple(d - 1 for d in dsum if d <= 4)
        if ax:
            y0 = y0.sum(axis=ax, keepdims=True)
    assert y1.shape[:4] == y0.shape
    tol = 1e4 * (np.finfo(np.float32).eps if prec == "single" else np.finfo(np.float64).eps) * np.abs(y0).max()
    if terp == "nearest" and prec == "single":
        return                                                                   # (rounding of a float32 index at .5: covered above statistically)
    assert np.abs(y1.reshape(y0.shape) - y0).max() <= 8 * tol


def test_wsinterpd_matches_the_oracle_with_phasor_extrapolation_and_sums():
    import torch
    from qups_amd.interpd import wsinterpd
    rng = np.random.default_rng(9)
    x = (rng.standard_normal((40, 3, 1, 2)) + 1j * rng.standard_normal((40, 3, 1, 2))).astype(np.complex128)
    t = -3 + 48 * rng.random((11, 3, 4, 1))                                      # some samples fall outside the record
    t[2, 1, 0, 0] = np.inf
    w = rng.random((1, 3, 4)) + 1j * rng.random((1, 3, 4))
    for terp in ("nearest", "linear", "cubic", "lanczos3"):
        for sdim, ev in ((None, np.nan), ([2], np.nan), ([2, 3], 0.0), (None, 2.5)):
            ref = O.wsinterpd(x, t, 1, w, sdim, terp, ev, 0.21j)
            y = _np(wsinterpd(torch.from_numpy(x), torch.from_numpy(t), 1, torch.from_numpy(w), sdim, terp, ev, 0.21j, prec="double"))
            assert y.shape == ref.shape
            assert np.array_equal(np.isnan(y), np.isnan(ref)), (terp, sdim)
            assert np.nanmax(np.abs(y - ref)) <= 1e-11, (terp, sdim, ev)


@pytest.mark.parametrize("prec", ["single", "double", "halfT"])
@pytest.mark.parametrize("terp", ["nearest", "linear", "cubic", "lanczos3"])
@pytest.mark.parametrize("M,wkind,ev", [(128, "none", 0.0), (33, "real", np.nan), (200, "complex", 0.0), (64, "real", 2.5), (31, "real", 0.0), (12, "real", 0.0)])
def test_wsinterpd_torch_order_record_summed_over_its_fastest_dimension(prec, terp, M, wkind, ev, monkeypatch):
    """A record in torch order (last dimension fastest) summed over that dimension: the lanes of a wave run along the sum (``wsinterpd_lanesum_kernel``;
    round 5 transposed the record first).  Against the float64 oracle, and against the transposed form (``QDAS_WS_NO_LANESUM``): the same terms, added in
    another order.  Delays out of the record, an infinite one and NaN ``extrapval`` (omitted by sums) included; ``M = 31``: two outputs per wave; ``M = 12``: below the kernel's 16 terms."""
    import torch
    from qups_amd.interpd import wsinterpd
    rng = np.random.default_rng(19)
    T, N = 300, 5
    dbl = prec == "double"
    ct = np.complex128 if dbl else np.complex64
    x = (rng.standard_normal((T, N, M)) + 1j * rng.standard_normal((T, N, M))).astype(ct)
    t = (np.arange(-3, T + 2, 2.5).reshape(-1, 1, 1) + rng.uniform(0, 3, (1, 1, M))).astype(np.float64 if dbl else np.float32)      # I x 1 x M: focusTx-like
    t[4, 0, M // 2] = np.inf
    w = None if wkind == "none" else rng.random((1, N, M)) + (1j * rng.random((1, N, M)) if wkind == "complex" else 0)
    if prec == "halfT":
        x = (x.real.astype(np.float16).astype(np.float32) + 1j * x.imag.astype(np.float16).astype(np.float32)).astype(np.complex64)
    wa = 1 if w is None else torch.from_numpy(w.astype(ct if wkind == "complex" else (np.float64 if dbl else np.float32)))
    om = 0.13j
    args = (torch.from_numpy(x), torch.from_numpy(t), 1, wa, [3], terp, ev, om)
    y = _np(wsinterpd(*args, prec=prec))
    ref = O.wsinterpd(x.astype(np.complex128), t.astype(np.float64), 1, 1 if w is None else w, [3], terp, ev, om)
    assert y.shape == ref.shape
    assert np.array_equal(np.isnan(y), np.isnan(ref))
    tol = 1e-11 if dbl else (2e-3 if prec == "halfT" else 3e-5)
    assert np.nanmax(np.abs(y - ref)) <= tol * max(1.0, np.nanmax(np.abs(ref))), (terp, M, wkind)
    monkeypatch.setenv("QDAS_WS_NO_LANESUM", "1")
    y2 = _np(wsinterpd(*args, prec=prec))
    assert np.array_equal(np.isnan(y), np.isnan(y2))
    assert np.nanmax(np.abs(y - y2)) <= tol * max(1.0, np.nanmax(np.abs(ref)))


@pytest.mark.parametrize("seed", range(24))
def test_wsinterpd_lane_sum_random_layouts(seed, monkeypatch):
    """Random index spaces of 3 - 5 dimensions whose ONE summed dimension is the fastest in memory (16 - 150 terms: two outputs per wave up to 32): broadcast dimensions in x / t / w, any
    memory order of the others, sampling dimension anywhere -- the lane-sum kernel against the float64 oracle and the one-output-per-lane kernel."""
    import torch
    from qups_amd.interpd import wsinterpd
    r = np.random.default_rng(4100 + seed)
    nd = int(r.integers(3, 6))
    dim = int(r.integers(1, nd))                                   # sampling dimension (1-based), not the last one
    S = int(r.choice([16, 17, 31, 32, 33, 64, 65, 100, 128, 150]))
    size = [int(r.integers(1, 6)) for _ in range(nd)]
    size[nd - 1] = S                                                # the summed, fastest dimension
    T, I = int(r.integers(20, 90)), int(r.integers(1, 40))
    xs, ts = list(size), list(size)
    xs[dim - 1], ts[dim - 1] = T, I
    for k in range(nd - 1):                                         # broadcast some of the other dimensions in x or in t
        if k != dim - 1 and r.integers(0, 3) == 0:
            (xs if r.integers(0, 2) else ts)[k] = 1
    if r.integers(0, 4) == 0:
        ts[nd - 1] = 1                                              # delays that do not depend on the summed index
    dbl = bool(r.integers(0, 3) == 0)
    ct, rt = (np.complex128, np.float64) if dbl else (np.complex64, np.float32)
    x = (r.standard_normal(xs) + 1j * r.standard_normal(xs)).astype(ct)
    t = r.uniform(-2, T + 1, ts).astype(rt)
    terp = str(r.choice(["nearest", "linear", "cubic", "lanczos3"]))
    wk = int(r.integers(0, 3))
    w = 1
    if wk:
        ws = [size[k] if r.integers(0, 2) else 1 for k in range(nd)]
        ws[dim - 1] = 1
        w = r.random(ws) + (1j * r.random(ws) if wk == 2 else 0)
    ev = float(r.choice([0.0, np.nan]))
    om = float(r.choice([0.0, 0.2])) * 1j
    perm = list(r.permutation(nd - 1)) + [nd - 1]                   # memory order of the leading dimensions (the last stays fastest)
    def lay(a):
        a = np.ascontiguousarray(np.transpose(a, perm))
        return torch.from_numpy(a).permute(*np.argsort(perm).tolist())
    wa = w if isinstance(w, int) else torch.from_numpy(np.asarray(w).astype(ct if wk == 2 else rt))
    args = (lay(x), lay(t), dim, wa, [nd], terp, ev, om)
    y = _np(wsinterpd(*args, prec="double" if dbl else "single"))
    ref = O.wsinterpd(x.astype(np.complex128), t.astype(np.float64), dim, w, [nd], terp, ev, om)
    assert y.shape == ref.shape, (y.shape, ref.shape)
    assert np.array_equal(np.isnan(y), np.isnan(ref))
    scale = max(1.0, float(np.nanmax(np.abs(ref))) if np.isfinite(ref).any() else 1.0)
    tol = (1e-11 if dbl else 5e-5) * scale
    if not (terp == "nearest" and not dbl):                          # (fp32 delays at .5 may round the other way)
        assert np.nanmax(np.abs(y - ref), initial=0.0) <= tol, (terp, size, dim)
    monkeypatch.setenv("QDAS_WS_NO_LANESUM", "1")
    y2 = _np(wsinterpd(*args, prec="double" if dbl else "single"))
    assert np.array_equal(np.isnan(y), np.isnan(y2))
    assert np.nanmax(np.abs(y - y2), initial=0.0) <= tol


def test_channeldata_sample_rectify_and_focusTx():
    """ChannelData.sample == oracle wsinterpd on (tau - t0) fs; rectifyt0 / rectifyDims; focusTx vs the oracle restatement and the
    physical check: plane-wave transmits synthesised from FSA data of a point target peak where a direct plane-wave simulation does"""
    import torch
    from qups_amd import ChannelData, Scan, Sequence, Transducer, UltrasoundSystem
    rng = np.random.default_rng(4)
    T, N, M = 200, 6, 6
    fs, c0 = 20e6, 1540.0
    x = (rng.standard_normal((T, N, M)) + 1j * rng.standard_normal((T, N, M))).astype(np.complex64)
    # -- sample: tau is I x N x 1, per-transmit t0
    t0 = (1e-6 + 2.37e-7 * np.arange(M)).reshape(1, 1, M)      # (not whole samples: the record edge would depend on rounding)
    chd = ChannelData(torch.from_numpy(x), t0, fs)
    tau = 2e-6 + 6e-6 * rng.random((50, N, 1))
    y = _np(chd.sample(tau, "cubic", fmod=1e6))
    ref = O.wsinterpd(x, (tau - t0) * fs, 1, 1, None, "cubic", 0.0, 2j * np.pi * 1e6 / fs)
    assert y.shape == ref.shape and rel_err(y, ref) <= 2e-5
    # -- rectifyDims / rectifyt0
    chd_p = ChannelData(torch.from_numpy(np.ascontiguousarray(x.transpose(1, 0, 2))), t0.transpose(1, 0, 2), fs, order="NTM")
    assert torch.equal(chd_p.rectifyDims().data, chd.data)
    r = chd.rectifyt0("cubic")
    rref, t0ref = O.rectifyt0(x, t0, fs, "cubic", index_dtype=np.float32)
    assert r.t0 == t0ref and tuple(r.data.shape) == rref.shape and rel_err(_np(r.data), rref) <= 2e-5
    # -- focusTx: FSA -> plane waves
    xdc = Transducer.linear(N, 0.3e-3)
    th = np.deg2rad([-8.0, 0.0, 8.0])
    seq = Sequence("PW", focus=np.stack([np.sin(th), 0 * th, np.cos(th)]), c0=c0, numPulse=3)
    us = UltrasoundSystem(xdc, Sequence("FSA", c0=c0), Scan.cartesian(np.linspace(-1e-3, 1e-3, 3), np.linspace(4e-3, 6e-3, 3)))
    chd0 = ChannelData(torch.from_numpy(x), 0.0, fs)
    z = us.focusTx(chd0, seq, interp="cubic")
    zref, t0z = O.focus_tx(x, 0.0, fs, O.sequence_delays("PW", xdc.positions(), seq.focus, c0), O.sequence_apodization("PW", N, 3), "cubic")
    assert abs(z.t0 - t0z) < 1e-15 and tuple(z.data.shape) == zref.shape
    assert rel_err(_np(z.data), zref) <= 2e-5
    assert np.allclose(seq.delays(xdc), O.sequence_delays("PW", xdc.positions(), seq.focus, c0))
    for typ, foc in (("FC", np.array([[0.0], [0.0], [20e-3]])), ("DV", np.array([[0.0], [0.0], [-5e-3]])), ("VS", np.array([[0.0, 1e-3], [0.0, 0.0], [10e-3, -4e-3]]))):
        assert np.allclose(Sequence(typ, focus=foc, c0=c0).delays(xdc), O.sequence_delays(typ, xdc.positions(), foc, c0))


def _shift_ref(x, shift, w, interp, To):
    """y[t', n, m', f] = sum_m w[m, m'] x(t' + shift[m, m'], n, m, f) through the float64 oracle sampler, one (m, m') pair at a time"""
    T, N, M = x.shape[:3]
    F = int(np.prod(x.shape[3:])) if x.ndim > 3 else 1
    Mo = shift.shape[1]
    xf = x.reshape(T, N, M, F).astype(np.complex128 if np.iscomplexobj(x) else np.float64)
    y = np.zeros((To, N, Mo, F), xf.dtype if not np.iscomplexobj(w) else np.complex128)
    tp = np.arange(To, dtype=np.float64)
    for m in range(M):
        for mo in range(Mo):
            if w[m, mo] == 0 or not np.isfinite(shift[m, mo]):
                continue
            pos = tp + shift[m, mo]
            for n in range(N):
                for f in range(F):
                    y[:, n, mo, f] += w[m, mo] * O.sample(xf[:, n, m, f], pos, interp)
    return y.reshape((To, N, Mo) + tuple(x.shape[3:]))


@pytest.mark.parametrize("interp", ["nearest", "linear", "cubic", "lanczos3", "cubic_dev"])
@pytest.mark.parametrize("dtype", ["complex64", "complex128", "float32", "float64"])
@pytest.mark.parametrize("dpp", [False, True])
def test_shift_sum_matches_the_oracle_and_the_general_kernel(interp, dtype, dpp, monkeypatch):
    """qdas_shift_sum (transmit synthesis: one offset per (element, synthesised transmit)) against the float64 oracle sampler and against
    the general single-delay kernel fed the materialised positions; offsets beyond both ends of the record, zero and complex weights, an
    odd block of synthesised transmits, frames, more and fewer output samples than the record holds"""
    import torch
    from qups_amd.interpd import shift_sum, wsinterpd
    if dpp:
        if dtype != "complex64":
            pytest.skip("lane-to-lane taps: fp32 complex data only")
        monkeypatch.setenv("QDAS_SS_DPP", "1")
    rng = np.random.default_rng(31)
    T, N, M, Mo, F = 700, 5, 11, 13, 2
    cplx = dtype.startswith("complex")
    x = rng.standard_normal((T, N, M, F)) + (1j * rng.standard_normal((T, N, M, F)) if cplx else 0)
    x = x.astype(dtype)
    shift = rng.uniform(-60, 60, (M, Mo))
    shift[0, 0], shift[1, 1], shift[2, 2], shift[3, 3] = -800.0, 800.0, 2.5, -3.5       # nothing in range twice; half-integers (nearest ties)
    shift[4, :] = np.round(shift[4, :])                                                  # integer offsets
    w = rng.uniform(0.2, 1, (M, Mo)) * (1 + (0.5j if cplx else 0))
    w[rng.random((M, Mo)) < 0.3] = 0
    w[5, :] = 0                                                                           # an element nobody uses
    dbl = dtype in ("complex128", "float64")
    tol = 1e-11 if dbl else 2e-5
    if not dbl:
        shift = shift.astype(np.float32).astype(np.float64)
    for To in (T, T + 150, 300):
        y = shift_sum(torch.from_numpy(x), shift, w, interp, To=To)
        ref = _shift_ref(x, shift, w, interp, To)
        assert tuple(y.shape) == ref.shape and y.dtype == getattr(torch, dtype)
        assert rel_err(_np(y), ref) <= tol, (To, interp)
    if interp == "nearest":                                                               # (discontinuous: a position rounded to fp32 may pick the neighbour)
        return
    # the general kernel on the same call (positions t' + shift as an array, summed over m)
    pos = (np.arange(T).reshape(T, 1, 1, 1) + shift.reshape(1, 1, M, Mo))
    z = wsinterpd(torch.from_numpy(x.reshape(T, N, M, 1, F)), torch.from_numpy(pos), 1, w.reshape(1, 1, M, Mo), [3], interp, 0.0)
    assert rel_err(_np(z).reshape(T, N, Mo, F), _np(shift_sum(torch.from_numpy(x), shift, w, interp))) <= (1e-11 if dbl else 2e-4)


def test_temporaries_of_interleaved_streams_do_not_collide():
    """the one-shot entries take their temporaries from an arena per (device, stream) (csrc/scratch.hip): calls of growing and shrinking size issued
    alternately on two streams -- each stream's arena regrown on the way -- give the bits of the same calls issued one by one"""
    import torch
    from qups_amd.interpd import shift_sum
    rng = np.random.default_rng(7)
    calls = []
    for k, (M, Mo) in enumerate([(5, 3), (40, 33), (7, 2), (64, 64), (3, 70), (64, 64)]):
        T, N = 300 + 17 * k, 4
        x = (rng.standard_normal((T, N, M)) + 1j * rng.standard_normal((T, N, M))).astype(np.complex64)
        shift = rng.uniform(-20, 20, (M, Mo)).astype(np.float32).astype(np.float64)
        w = rng.uniform(0.2, 1, (M, Mo))
        calls.append((torch.from_numpy(x).cuda(), shift, w))
    ref = [shift_sum(x, sh, w, "cubic").clone() for x, sh, w in calls]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    out = [None] * len(calls)
    for rep in range(3):
        for k, (x, sh, w) in enumerate(calls):
            with torch.cuda.stream(s1 if (k + rep) % 2 == 0 else s2):
                out[k] = shift_sum(x, sh, w, "cubic")
        torch.cuda.synchronize()
        for k in range(len(calls)):
            assert torch.equal(out[k], ref[k]), (rep, k)


def test_shift_sum_offsets_too_far_apart_for_one_window_and_no_weights():
    """offsets spread over more than a staged window holds (a window per synthesised transmit), weights omitted (ones), one receiver"""
    import torch
    from qups_amd.interpd import shift_sum
    rng = np.random.default_rng(32)
    T, N, M, Mo = 9000, 1, 3, 9
    x = (rng.standard_normal((T, N, M)) + 1j * rng.standard_normal((T, N, M))).astype(np.complex64)
    shift = rng.uniform(-4000, 4000, (M, Mo)).astype(np.float32).astype(np.float64)
    y = shift_sum(torch.from_numpy(x), shift, None, "cubic")
    ref = _shift_ref(x, shift, np.ones((M, Mo)), "cubic", T)
    assert rel_err(_np(y), ref) <= 2e-5
    with pytest.raises(Exception):
        shift_sum(torch.from_numpy(x.real.copy()), shift, np.ones((M, Mo)) * 1j, "cubic")     # real data take real weights


@pytest.mark.parametrize("dtype,interp,Mo", [("complex64", "cubic", 8), ("complex64", "linear", 5), ("float32", "lanczos3", 8), ("complex128", "cubic", 3),
                                             ("complex64", "nearest", 16)])
@pytest.mark.parametrize("dpp", [False, True])
def test_shift_sum_zero_tail_that_is_not_stored(dtype, interp, Mo, dpp, monkeypatch):
    """``tpad`` (``include/qdas.h``): the record counts as followed by zeros -- bit for bit what the same call returns on a zero-padded copy
    (``focusTx`` no longer makes that copy), on the paths with and without support masks; and ``focusTx`` itself against its padded form."""
    import torch
    from qups_amd.interpd import shift_sum
    if dpp:
        monkeypatch.setenv("QDAS_SS_DPP", "1")
    rng = np.random.default_rng(77)
    T, N, M, pad = 1500, 3, 6, 211
    x = rng.standard_normal((T, N, M)).astype(np.float32)
    if dtype.startswith("complex"):
        x = x + 1j * rng.standard_normal((T, N, M)).astype(np.float32)
    x = x.astype(dtype)
    shift = -rng.uniform(0, pad, (M, Mo))                                              # (focusTx: delays shifted to be non-negative)
    w = rng.uniform(0.5, 1, (M, Mo))
    xp = np.concatenate([x, np.zeros((pad, N, M), x.dtype)], 0)
    a = shift_sum(torch.from_numpy(x), shift, w, interp, tpad=pad)
    b = shift_sum(torch.from_numpy(xp), shift, w, interp)
    assert a.shape == b.shape == (T + pad, N, Mo)
    assert torch.equal(a, b)
    a = shift_sum(torch.from_numpy(x), shift, w, interp, To=T + 40, tpad=pad)          # fewer outputs than the padded length
    assert torch.equal(a, b[:T + 40])
    with pytest.raises(Exception):
        shift_sum(torch.from_numpy(x), shift, w, interp, tpad=-1)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("QDAS_SHIFT_FUZZ", "32"))))
def test_shift_sum_random_configuration(seed, monkeypatch):
    """random record / output lengths (around the 256 x {2,3,4}-sample blocks), element and transmit counts (ragged blocks of 8), frames, types,
    interpolators, offsets (clustered: one window per element; scattered: a window per transmit), sparse real / complex weights"""
    import torch
    from qups_amd.interpd import shift_sum
    r = np.random.default_rng(12000 + seed)
    if seed % 2:                                        # complex64 data: taps passed from lane to lane (groups of 65 - K outputs) whatever the block count says
        monkeypatch.setenv("QDAS_SS_DPP", "1")
    T = int(r.choice([5, 64, 255, 256, 257, 700, 1023, 1025, 2100, 5000]))
    To = int(r.choice([T, T, max(1, T // 2), T + 37, 1]))
    N, M, Mo, F = int(r.integers(1, 4)), int(r.choice([1, 2, 7, 12])), int(r.choice([1, 3, 8, 9, 17])), int(r.choice([1, 1, 2]))
    dtype = str(r.choice(["complex64", "complex64", "complex128", "float32", "float64"]))
    interp = str(r.choice(["nearest", "linear", "cubic", "lanczos3", "cubic_dev"]))
    cplx = dtype.startswith("complex")
    x = (r.standard_normal((T, N, M, F)) + (1j * r.standard_normal((T, N, M, F)) if cplx else 0)).astype(dtype)
    spread = float(r.choice([3.0, 40.0, 600.0, 6000.0]))
    shift = r.uniform(-spread, spread, (M, Mo)) + r.choice([0.0, -T / 3, T / 3])
    if r.integers(0, 3) == 0:
        shift = np.round(shift * 2) / 2                                                  # integers and half-integers
    dbl = dtype in ("complex128", "float64")
    if not dbl:
        shift = shift.astype(np.float32).astype(np.float64)
    wkind = int(r.integers(0, 4))
    w = None if wkind == 0 else r.uniform(0.1, 1, (M, Mo)) * ((1 + 0.3j) if (wkind == 3 and cplx) else 1)
    if w is not None:
        w[r.random((M, Mo)) < float(r.choice([0.0, 0.4, 0.9]))] = 0
    y = shift_sum(torch.from_numpy(x), shift, w, interp, To=To)
    ref = _shift_ref(x, shift, np.ones((M, Mo)) if w is None else w, interp, To)
    scale = max(np.abs(ref).max(), 1e-30)
    assert tuple(y.shape) == ref.shape
    assert np.abs(_np(y) - ref).max() / scale <= (1e-11 if dbl else 3e-5) or np.abs(ref).max() == 0 and np.abs(_np(y)).max() == 0, (seed, T, To, N, M, Mo, F, dtype, interp, spread, wkind)
