"""GPU tests against the committed golden fixtures (tests/golden/*.npz) and of the caller-level mirrors
(UltrasoundSystem.DAS / bfDAS, sample2sep / wsinterpd2) -- everything goes through the C ABI of libqdas.so."""
import os

import numpy as np
import pytest

from tests.cases import cinv_f32, make_case, rel_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name))


def _np(t):
    import torch
    return t.to(torch.complex128).cpu().numpy() if t.is_complex() else t.cpu().numpy()


@pytest.mark.parametrize("terp", ["nearest", "linear", "cubic", "lanczos3", "cubic_dev"])
@pytest.mark.parametrize("prec", ["single", "double"])
def test_f1_interptest_fixture_wsinterpd2(terp, prec):
    """the reference's own kernel test (test/interpTest.m:97-143) on its closed-form fixture"""
    import torch
    from qups_amd import wsinterpd2
    g = gold("f1_interptest.npz")
    tau, x0 = g["tau"], g["x0"]                               # I x N x M ; T x N x F
    I, N, M = tau.shape
    F = x0.shape[2]
    i = np.arange(I)[:, None, None]
    n = np.arange(N)[None, :, None]
    m = np.arange(M)[None, None, :]
    T = x0.shape[0]
    t1 = 4 + (T - 8) * ((1 + i) / I * 1 / N * (1 + m) / M)    # I x 1 x M   (interpTest.m:38)
    t2 = 4 + (T - 8) * 1 / I * (1 + n) / N                    # 1 x N x 1   (interpTest.m:39)
    x = torch.from_numpy(x0[:, :, None, :].astype(np.complex64 if prec == "single" else np.complex128))   # T x N x 1 x F
    tol = (1e4 * np.finfo(np.float32 if prec == "single" else np.float64).eps) * 50     # interpTest.m:126 (x50: tau*fs in fp32)
    y = _np(wsinterpd2(x, t1, t2, 1, 1, None, terp, 0, prec=prec))                       # I x N x M x F
    ref = g["y_" + terp]
    assert y.shape == ref.shape
    if terp != "nearest":
        assert np.abs(y - ref).max() <= tol * np.abs(ref).max() * (40 if prec == "single" else 1)
    else:
        assert np.mean(np.abs(y - ref) > 1e-5) < 0.01
    w = np.random.default_rng(3).uniform(size=(1, 1, M)).astype(np.float32)              # weights over dim 3, summed over [2 3]
    ys = _np(wsinterpd2(x, t1, t2, 1, w, [2, 3], terp, 0, prec=prec))
    if terp != "nearest":
        assert rel_err(ys[:, 0, 0, :], (w[..., None] * ref).sum((1, 2))) <= (2e-5 if prec == "single" else 1e-12)


@pytest.mark.parametrize("seq", ["FSA", "PW", "FC"])
def test_f2_psf_DAS_and_bfDAS(seq):
    """UltrasoundSystem.DAS and bfDAS reproduce the frozen PSF image and the BFTest criterion (test/BFTest.m:306-316)"""
    import torch
    from qups_amd import ChannelData, Scan, Sequence, Transducer, UltrasoundSystem
    g = gold("f2_psf.npz")
    k = lambda s: g[seq + "_" + s]
    xdc = Transducer(k("Pr"), np.stack([0 * k("Pr")[0], 0 * k("Pr")[0], 1 + 0 * k("Pr")[0]]))
    focus = k("Nv") if seq == "PW" else k("Pv")
    us = UltrasoundSystem(xdc, Sequence(seq, focus=focus, c0=float(k("c"))), Scan(k("Pi")))
    chd = ChannelData(torch.from_numpy(k("x")), float(k("t0")), float(k("fs")))
    ref = k("y")
    b = _np(us.DAS(chd, interp="cubic"))[..., 0, 0] if False else _np(us.DAS(chd, interp="cubic")).reshape(ref.shape)
    assert rel_err(b, ref) <= (1e-4 if seq == "FC" else 2e-5)
    iz, ix, _ = np.unravel_index(np.argmax(np.abs(b)), b.shape)
    assert np.count_nonzero(b) and abs(k("dx")[ix]) <= 1.1e-3 and abs(k("dz")[iz]) <= 1.1e-3
    b2 = _np(us.bfDAS(chd, interp="cubic")).reshape(ref.shape)                # split-delay flavour, fp32 tables
    assert rel_err(b2, ref) <= 5e-4
    iz, ix, _ = np.unravel_index(np.argmax(np.abs(b2)), b2.shape)
    assert abs(k("dx")[ix]) <= 1.1e-3 and abs(k("dz")[iz]) <= 1.1e-3


def test_f3_modes_golden():
    import torch
    from qups_amd import das_spec
    g = gold("f3_modes.npz")
    args = (g["Pi"], g["Pr"], g["Pv"], g["Nv"], torch.from_numpy(g["x"]), float(g["t0"]), float(g["fs"]), float(g["c"]))
    for fun in ("DAS", "SYN", "MUL", "BF"):
        y = _np(das_spec(fun, *args, "plane-waves", "interp", "linear", "apod", g["a1"], "apod", g["a2"]))
        assert y.shape == g["y_" + fun].shape
        assert rel_err(y, g["y_" + fun]) <= 1e-4
    y = _np(das_spec("DAS", *args, "plane-waves", "interp", "linear", "modulation", float(np.float32(3e6))))
    assert rel_err(y, g["y_DAS_fmod"]) <= 2e-4


def test_f4_edges_golden_lut_kernel():
    import torch
    from qups_amd import das_lut
    g = gold("f4_edges.npz")
    x, s = g["x"], g["s"]
    for terp in ("nearest", "linear", "cubic", "lanczos3"):
        y = _np(das_lut(torch.from_numpy(x.reshape(-1, 1, 1)), s.reshape(-1, 1), np.zeros((len(s), 1)), interp=terp, prec="double"))
        ref = g["y_" + terp]
        assert np.array_equal(y.reshape(-1) == 0, np.nan_to_num(ref) == 0), terp     # exactly zero out of support
        assert np.abs(y.reshape(-1) - np.nan_to_num(ref)).max() <= 1e-12


@pytest.mark.parametrize("keep", [(False, False), (True, False), (False, True), (True, True)])
def test_das_lut_matches_oracle(keep):
    import torch
    from oracle import das_oracle as O
    from qups_amd import sample2sep
    case = make_case(seq="FSA", interp="cubic", seed=21, N=6, I1=40, I2=5)
    dv, dr = O.tx_rx_distances(case["Pi"], case["Pr"], case["Pv"], case["Nv"], True, True)
    c = cinv_f32(case["c"])
    tau_tx = (dv[:, :, :, 0, :] / c)
    tau_rx = (dr[:, :, :, :, 0] / c)
    rng = np.random.default_rng(0)
    w = rng.uniform(0.2, 1, (40, 5, 1, 6, 1)).astype(np.float32).astype(np.float64)
    keep_rx, keep_tx = keep
    fm = float(np.float32(2e6))
    ref = O.das_lut(case["x"], tau_rx, tau_tx, case["t0"], case["fs"], interp="cubic", apod=(w,), fmod=fm, keep_rx=keep_rx, keep_tx=keep_tx)
    sd = set() if keep_rx else {"rx"}
    sd |= set() if keep_tx else {"tx"}
    y = _np(sample2sep(torch.from_numpy(case["x"]), case["t0"], case["fs"], tau_rx, tau_tx, "cubic", w, sd, fm, prec="double"))
    assert y.shape == ref.shape and rel_err(y, ref) <= 1e-9
    y32 = _np(sample2sep(torch.from_numpy(case["x"]), case["t0"], case["fs"], tau_rx, tau_tx, "cubic", w, sd, fm, prec="single"))
    assert rel_err(y32, ref) <= 5e-4


@pytest.mark.parametrize("interp", ["nearest", "linear", "cubic", "lanczos3"])
@pytest.mark.parametrize("seq,tpose,wtab,fm", [("FSA", False, False, 0.0), ("PW", True, True, 2e6), ("DV", False, True, 0.0), ("FC", False, False, 2e6)])
def test_das_lut_full_sum_runs_the_fused_kernel(interp, seq, tpose, wtab, fm, monkeypatch):
    """fp32 full-sum calls with the image shape known go through the tiled kernel with table-driven delays (launch configuration
    10): must agree with the oracle and with the one-thread-per-pixel kernel (QDAS_LUT_GENERIC=1), also with an N x M weight
    table, remodulation, transposed data, a ragged last tile and delays that leave the record"""
    import torch
    from oracle import das_oracle as O
    from qups_amd import das_lut
    case = make_case(seq=seq, interp=interp, seed=23, N=16, M=None if seq == "FSA" else 9, I1=150, I2=21, data="noise")
    N, M = case["N"], case["M"]
    dv, dr = O.tx_rx_distances(case["Pi"], case["Pr"], case["Pv"], case["Nv"], case["VS"], case["DV"])
    c = cinv_f32(case["c"])
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    tau_tx = f32((dv[:, :, :, 0, :] / c - case["t0"]) * case["fs"])[:, :, 0]           # I1 x I2 x M, in samples, fp32-representable
    tau_rx = f32(dr[:, :, :, :, 0] / c * case["fs"])[:, :, 0]                           # I1 x I2 x N
    rng = np.random.default_rng(1)
    w = None
    if wtab:
        w = (rng.uniform(0.2, 1, (1, 1, N, M)) + 1j * rng.uniform(-0.3, 0.3, (1, 1, N, M))).astype(np.complex64)
    omega = 2 * np.pi * fm / case["fs"]
    x = case["x"]
    xs = np.ascontiguousarray(np.swapaxes(x, 1, 2)) if tpose else x
    ref = O.das_lut(x, tau_rx[:, :, None] / case["fs"], tau_tx[:, :, None] / case["fs"], 0.0, case["fs"], interp=interp,
                    apod=(() if w is None else (w.reshape(1, 1, 1, N, M).astype(np.complex128),)), fmod=fm)
    ref = np.asarray(ref).reshape(150, 21)
    run = lambda: _np(das_lut(torch.from_numpy(xs), tau_rx, tau_tx, interp=interp, w=w, omega=omega, prec="single", tpose=tpose)).reshape(150, 21)
    monkeypatch.delenv("QDAS_LUT_GENERIC", raising=False)
    a = run()
    monkeypatch.setenv("QDAS_LUT_GENERIC", "1")
    b = run()
    den = np.abs(ref).max()
    if interp == "nearest":                                   # a rounding at the step swaps one sample of a pixel: rare
        assert (np.abs(a - ref) / den > 1e-4).mean() <= 0.05 and (np.abs(a - b) / den > 1e-4).mean() <= 0.05
    else:
        assert rel_err(a, ref) <= 1e-4 and rel_err(a, b) <= 1e-4
    assert not np.array_equal(a, b)                           # two different kernels did run
    assert np.all(a[np.abs(ref) == 0] == 0)


@pytest.mark.parametrize("interp,seq,fm,I2,M", [("cubic", "PW", 0.0, 40, 32), ("lanczos3", "FSA", 0.0, 40, 32), ("linear", "PW", 1.5e6, 40, 32), ("cubic", "DV", 0.0, 40, 32),
                                                 ("cubic", "PW", 0.0, 41, 32), ("cubic", "PW", 0.0, 40, 8)], ids=["pw", "fsa-lanczos", "pw-fmod", "dv", "odd-columns", "few-transmits-roles-swapped"])
def test_das_lut_mirror_symmetric_tables_take_the_mirror_mode(interp, seq, fm, I2, M, monkeypatch):
    """delay tables that are their own lateral mirror images (a centred scan, a symmetric probe and sequence: what bfDASLUT builds for the BASELINE
    configurations) run the two-window-set build -- a pixel and its image share tap index and weights, as geometry-driven plans do -- and agree with the oracle and
    with the general table-driven kernel (QDAS_LUT_NO_MIRROR=1).  The symmetry is checked PER CALL, bit for bit: ONE table entry one ulp off must NOT take the mode
    (and still be exact)."""
    import torch
    from oracle import das_oracle as O
    from qups_amd import das_lut
    case = make_case(seq=seq, interp=interp, seed=29, N=32, M=M, I1=140, I2=I2, data="noise")
    N, M = case["N"], case["M"]
    dv, dr = O.tx_rx_distances(case["Pi"], case["Pr"], case["Pv"], case["Nv"], case["VS"], case["DV"])
    c = cinv_f32(case["c"])
    f32 = lambda a: np.asarray(a, np.float32)
    tau_tx = f32((dv[:, :, :, 0, :] / c - case["t0"]) * case["fs"])[:, :, 0]           # I1 x I2 x M, samples
    tau_rx = f32(dr[:, :, :, :, 0] / c * case["fs"])[:, :, 0]                           # I1 x I2 x N
    # make the tables EXACTLY symmetric (the fp32 geometry of make_case is symmetric to rounding only): second half := mirror image of the first
    h = tau_rx.shape[1] // 2
    tau_rx[:, -h:, :] = tau_rx[:, :h, :][:, ::-1, ::-1]
    tau_tx[:, -h:, :] = tau_tx[:, :h, :][:, ::-1, ::-1]
    if I2 % 2:                                                # the centre column is its own image: its rows must be symmetric in the element index
        tau_rx[:, h, :] = 0.5 * (tau_rx[:, h, :] + tau_rx[:, h, ::-1])
        tau_tx[:, h, :] = 0.5 * (tau_tx[:, h, :] + tau_tx[:, h, ::-1])
    omega = 2 * np.pi * fm / case["fs"]
    x = case["x"]
    ref = np.asarray(O.das_lut(x, tau_rx.astype(np.float64)[:, :, None] / case["fs"], tau_tx.astype(np.float64)[:, :, None] / case["fs"], 0.0, case["fs"], interp=interp, fmod=fm)).reshape(140, I2)
    run = lambda trx, ttx: _np(das_lut(torch.from_numpy(x), trx, ttx, interp=interp, omega=omega, prec="single")).reshape(140, I2)
    monkeypatch.delenv("QDAS_LUT_NO_MIRROR", raising=False)
    a = run(tau_rx, tau_tx)
    if "mirror" not in das_lut.last_kernel:                   # (no hiprtc on this box: the mode exists as a specialised build only)
        pytest.skip("no mirror build: " + das_lut.last_kernel)
    assert das_lut.last_kernel.startswith("tiled,mirror") and "[jit " in das_lut.last_kernel
    monkeypatch.setenv("QDAS_LUT_NO_MIRROR", "1")
    b = run(tau_rx, tau_tx)
    assert das_lut.last_kernel == "tiled"
    monkeypatch.delenv("QDAS_LUT_NO_MIRROR")
    assert rel_err(a, ref) <= 2e-5 and rel_err(a, b) <= 2e-5, (rel_err(a, ref), rel_err(a, b))      # (two summation orders of fp32 sums)
    assert not np.array_equal(a, b)                           # two different kernels did run
    # one entry of one table one ulp off: not symmetric -> the general kernel, same image as ITS oracle
    t2 = tau_rx.copy()
    t2[17, 3, 5] = np.nextafter(t2[17, 3, 5], np.float32(np.inf))
    c2 = run(t2, tau_tx)
    assert das_lut.last_kernel == "tiled"
    assert rel_err(c2, b) <= 1e-5
    t3 = tau_tx.copy()
    t3[100, I2 - 1, M - 1] = np.nextafter(t3[100, I2 - 1, M - 1], np.float32(-np.inf))
    run(tau_rx, t3)
    assert das_lut.last_kernel == "tiled"


@pytest.mark.parametrize("prec,real,keep_rx", [("single", True, False), ("single", False, False), ("single", True, True), ("halfT", True, False), ("halfT", False, False)])
def test_das_lut_receive_apodization_array_on_the_fused_kernel(prec, real, keep_rx, monkeypatch):
    """the usual bfDASLUT call: a pixel x receiver apodization (I1 x I2 x N, real or complex, data precision) -- applied per stage by
    the fused kernel like the plans' arrays; also with the receive dimension kept"""
    import torch
    from oracle import das_oracle as O
    from qups_amd import das_lut
    case = make_case(seq="DV", interp="cubic", seed=31, N=16, M=9, I1=130, I2=19, data="noise")
    N, M = case["N"], case["M"]
    dv, dr = O.tx_rx_distances(case["Pi"], case["Pr"], case["Pv"], case["Nv"], case["VS"], case["DV"])
    c = cinv_f32(case["c"])
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    tau_tx = f32((dv[:, :, :, 0, :] / c - case["t0"]) * case["fs"])[:, :, 0]
    tau_rx = f32(dr[:, :, :, :, 0] / c * case["fs"])[:, :, 0]
    rng = np.random.default_rng(2)
    q = (lambda a: a.astype(np.float16).astype(np.float64)) if prec == "halfT" else (lambda a: a.astype(np.float32).astype(np.float64))
    w = q(rng.uniform(0, 1, (130, 19, N, 1)) * (rng.uniform(0, 1, (130, 19, N, 1)) > 0.3))
    if not real:
        w = w + 1j * q(rng.uniform(-0.3, 0.3, (130, 19, N, 1)))
    x = case["x"]
    if prec == "halfT":
        x = (x.real.astype(np.float16).astype(np.float32) + 1j * x.imag.astype(np.float16).astype(np.float32)).astype(np.complex64)
    ref = np.asarray(O.das_lut(x, tau_rx[:, :, None] / case["fs"], tau_tx[:, :, None] / case["fs"], 0.0, case["fs"], interp="cubic",
                               apod=(w.reshape(130, 19, 1, N, 1).astype(np.complex128),), keep_rx=keep_rx)).reshape(130, 19, -1)
    wt = torch.from_numpy(w.astype(np.float32) if real else w.astype(np.complex64))
    run = lambda: _np(das_lut(torch.from_numpy(x), tau_rx, tau_tx, interp="cubic", w=wt, prec=prec, keep_rx=keep_rx).to(torch.complex64)).reshape(130, 19, -1)
    monkeypatch.delenv("QDAS_LUT_GENERIC", raising=False)
    a = run()
    monkeypatch.setenv("QDAS_LUT_GENERIC", "1")
    b = run()
    tol = 3e-3 if prec == "halfT" else 1e-4
    assert a.shape == ref.shape and rel_err(a, ref) <= tol and rel_err(a, b) <= tol
    assert not np.array_equal(a, b)


@pytest.mark.parametrize("prec,wkind", [("single", "pixm"), ("halfT", "pixm"), ("single", "none")])
def test_das_lut_with_swapped_tables(prec, wkind, monkeypatch):
    """the split-delay flavour with the two tables trading places in a full sum: a weight per (pixel, TRANSMIT) -- I1 x I2 x 1 x M, scanline-style
    transmit apodization -- applied per stage with the transmit as the stage element; and, without weights, few transmits against many
    receivers (fewer, fuller stages).  Oracle + the one-thread-per-pixel kernel."""
    import torch
    from oracle import das_oracle as O
    from qups_amd import das_lut
    case = make_case(seq="FC", interp="cubic", seed=37, N=40, M=6, I1=130, I2=24, data="noise")
    N, M = case["N"], case["M"]
    dv, dr = O.tx_rx_distances(case["Pi"], case["Pr"], case["Pv"], case["Nv"], case["VS"], case["DV"])
    c = cinv_f32(case["c"])
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    tau_tx = f32((dv[:, :, :, 0, :] / c - case["t0"]) * case["fs"])[:, :, 0]
    tau_rx = f32(dr[:, :, :, :, 0] / c * case["fs"])[:, :, 0]
    rng = np.random.default_rng(6)
    q = (lambda a: a.astype(np.float16).astype(np.float64)) if prec == "halfT" else (lambda a: a.astype(np.float32).astype(np.float64))
    x = case["x"]
    if prec == "halfT":
        x = (x.real.astype(np.float16).astype(np.float32) + 1j * x.imag.astype(np.float16).astype(np.float32)).astype(np.complex64)
    w, apod = None, ()
    if wkind == "pixm":
        w = q(rng.uniform(0, 1, (130, 24, 1, M)) * (rng.uniform(0, 1, (130, 24, 1, M)) > 0.4))
        apod = (w.reshape(130, 24, 1, 1, M).astype(np.complex128),)
    ref = np.asarray(O.das_lut(x, tau_rx[:, :, None] / case["fs"], tau_tx[:, :, None] / case["fs"], 0.0, case["fs"], interp="cubic", apod=apod)).reshape(130, 24)
    wt = None if w is None else torch.from_numpy(w.astype(np.float32))
    run = lambda: _np(das_lut(torch.from_numpy(x), tau_rx, tau_tx, interp="cubic", w=wt, prec=prec).to(torch.complex64)).reshape(130, 24)
    monkeypatch.delenv("QDAS_LUT_GENERIC", raising=False)
    a = run()
    monkeypatch.setenv("QDAS_LUT_GENERIC", "1")
    b = run()
    tol = 3e-3 if prec == "halfT" else 1e-4
    assert rel_err(a, ref) <= tol and rel_err(a, b) <= tol
    assert not np.array_equal(a, b)                              # two different kernels did run


@pytest.mark.parametrize("keep", ["rx", "tx"])
@pytest.mark.parametrize("seq,interp,tpose,fm", [("FSA", "cubic", False, 0.0), ("PW", "lanczos3", True, 2e6), ("FC", "linear", False, 0.0)])
def test_das_lut_one_kept_dimension_on_the_fused_kernel(keep, seq, interp, tpose, fm, monkeypatch):
    """keep the receive or the transmit dimension (sum over the other one): the fused kernel's plane-per-stage mode, with the two
    tables trading places for the transmit dimension; oracle + one-thread-per-output kernel"""
    import torch
    from oracle import das_oracle as O
    from qups_amd import das_lut
    case = make_case(seq=seq, interp=interp, seed=27, N=16, M=None if seq == "FSA" else 9, I1=130, I2=19, data="noise")
    N, M = case["N"], case["M"]
    dv, dr = O.tx_rx_distances(case["Pi"], case["Pr"], case["Pv"], case["Nv"], case["VS"], case["DV"])
    c = cinv_f32(case["c"])
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    tau_tx = f32((dv[:, :, :, 0, :] / c - case["t0"]) * case["fs"])[:, :, 0]
    tau_rx = f32(dr[:, :, :, :, 0] / c * case["fs"])[:, :, 0]
    x = case["x"]
    xs = np.ascontiguousarray(np.swapaxes(x, 1, 2)) if tpose else x
    kr, kt = keep == "rx", keep == "tx"
    ref = np.asarray(O.das_lut(x, tau_rx[:, :, None] / case["fs"], tau_tx[:, :, None] / case["fs"], 0.0, case["fs"], interp=interp, fmod=fm,
                               keep_rx=kr, keep_tx=kt)).reshape(130, 19, N if kr else M)
    run = lambda: _np(das_lut(torch.from_numpy(xs), tau_rx, tau_tx, interp=interp, omega=2 * np.pi * fm / case["fs"], prec="single", tpose=tpose,
                              keep_rx=kr, keep_tx=kt)).reshape(130, 19, -1)
    monkeypatch.delenv("QDAS_LUT_GENERIC", raising=False)
    a = run()
    monkeypatch.setenv("QDAS_LUT_GENERIC", "1")
    b = run()
    assert a.shape == ref.shape and rel_err(a, ref) <= 1e-4 and rel_err(a, b) <= 1e-4
    assert not np.array_equal(a, b)


@pytest.mark.parametrize("interp", ["nearest", "linear", "cubic", "lanczos3"])
def test_das_lut_half_precision_data_on_the_fused_kernel(interp, monkeypatch):
    """fp16 channel data + fp32 delay tables (launch configuration 11) against the oracle on the fp16-rounded data and against the
    one-thread-per-output kernel"""
    import torch
    from oracle import das_oracle as O
    from qups_amd import das_lut
    case = make_case(seq="DV", interp=interp, seed=29, N=16, M=9, I1=150, I2=21, data="noise")
    N, M = case["N"], case["M"]
    dv, dr = O.tx_rx_distances(case["Pi"], case["Pr"], case["Pv"], case["Nv"], case["VS"], case["DV"])
    c = cinv_f32(case["c"])
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    tau_tx = f32((dv[:, :, :, 0, :] / c - case["t0"]) * case["fs"])[:, :, 0]
    tau_rx = f32(dr[:, :, :, :, 0] / c * case["fs"])[:, :, 0]
    xh = (case["x"].real.astype(np.float16).astype(np.float32) + 1j * case["x"].imag.astype(np.float16).astype(np.float32)).astype(np.complex64)
    ref = np.asarray(O.das_lut(xh, tau_rx[:, :, None] / case["fs"], tau_tx[:, :, None] / case["fs"], 0.0, case["fs"], interp=interp)).reshape(150, 21)
    run = lambda: _np(das_lut(torch.from_numpy(xh), tau_rx, tau_tx, interp=interp, prec="halfT").to(torch.complex64)).reshape(150, 21)
    monkeypatch.delenv("QDAS_LUT_GENERIC", raising=False)
    a = run()
    monkeypatch.setenv("QDAS_LUT_GENERIC", "1")
    b = run()
    den = np.abs(ref).max()
    if interp == "nearest":
        assert (np.abs(a - ref) / den > 3e-3).mean() <= 0.05
    else:
        assert rel_err(a, ref) <= 3e-3 and rel_err(a, b) <= 3e-3      # fp16 output rounding
    assert not np.array_equal(a, b)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("QDAS_LUT_FUZZ", "32"))))
def test_das_lut_random_configuration(seed):
    """random image shapes (ragged tiles, small images -> aperture split), apertures, interpolators, sequences, weight tables,
    remodulation, transposed data, and -- every fourth case -- a non-finite delay, which sends the call to the one-thread-per-output
    kernel (the reference skips such samples, src/interpd.cu:390)"""
    import torch
    from oracle import das_oracle as O
    from qups_amd import das_lut
    r = np.random.default_rng(4000 + seed)
    seq = str(r.choice(["FSA", "PW", "DV", "FC"]))
    interp = str(r.choice(["nearest", "linear", "cubic", "lanczos3"]))
    N = int(r.choice([1, 2, 7, 16, 33]))
    M = int(r.choice([1, 3, 16, 32, 40]))
    I1, I2 = int(r.integers(1, 200)), int(r.integers(1, 40))
    case = make_case(seq=seq, interp=interp, seed=seed, N=N, M=M, I1=I1, I2=I2, zlim=(4e-3, 4e-3 + max(I1, 2) * 0.1e-3), xspan=2e-3,
                     data=str(r.choice(["noise", "targets"])))
    N, M = case["N"], case["M"]
    dv, dr = O.tx_rx_distances(case["Pi"], case["Pr"], case["Pv"], case["Nv"], case["VS"], case["DV"])
    c = cinv_f32(case["c"])
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    tau_tx = f32((dv[:, :, :, 0, :] / c - case["t0"]) * case["fs"])[:, :, 0]
    tau_rx = f32(dr[:, :, :, :, 0] / c * case["fs"])[:, :, 0]
    if seed % 4 == 3:
        tau_rx[r.integers(0, I1), r.integers(0, I2), r.integers(0, N)] = np.nan
    w = None
    if r.integers(0, 2):
        w = (r.uniform(0.2, 1, (1, 1, N, M)) + 1j * r.uniform(-0.3, 0.3, (1, 1, N, M))).astype(np.complex64)
    fm = float(r.choice([0.0, 2e6]))
    tpose = bool(r.integers(0, 2))
    x = case["x"]
    xs = np.ascontiguousarray(np.swapaxes(x, 1, 2)) if tpose else x
    ref = np.asarray(O.das_lut(x, tau_rx[:, :, None] / case["fs"], tau_tx[:, :, None] / case["fs"], 0.0, case["fs"], interp=interp,
                               apod=(() if w is None else (w.reshape(1, 1, 1, N, M).astype(np.complex128),)), fmod=fm)).reshape(I1, I2)
    y = _np(das_lut(torch.from_numpy(xs), tau_rx, tau_tx, interp=interp, w=w, omega=2 * np.pi * fm / case["fs"], prec="single", tpose=tpose)).reshape(I1, I2)
    den = np.abs(ref).max()
    if den == 0:
        assert np.abs(y).max() == 0
        return
    bad = np.abs(y - ref) / den > 1e-4
    if interp == "nearest":
        assert bad.mean() <= 0.05, (seed, seq, interp, N, M, I1, I2)
    else:
        assert bad.sum() <= max(2, bad.size // 500), (seed, seq, interp, N, M, I1, I2, float((np.abs(y - ref) / den).max()))   # edge-rule flips are rare
    assert np.all(y[np.abs(ref) == 0] == 0)


def test_us_DAS_keep_dims_and_frames_layout():
    import torch
    from qups_amd import ChannelData, Scan, Sequence, Transducer, UltrasoundSystem
    case = make_case(seq="PW", interp="linear", seed=22, N=6, M=4, I1=30, I2=5)
    xdc = Transducer(case["Pr"], np.stack([0 * case["Pr"][0], 0 * case["Pr"][0], 1 + 0 * case["Pr"][0]]))
    us = UltrasoundSystem(xdc, Sequence("PW", focus=case["Nv"], c0=case["c"]), Scan(case["Pi"]))
    xf = np.stack([case["x"], 2 * case["x"]], axis=3)
    chd = ChannelData(torch.from_numpy(xf), case["t0"], case["fs"])
    b = us.DAS(chd, interp="linear", keep_rx=True, keep_tx=True)
    assert tuple(b.shape) == (30, 5, 1, 2, 6, 4)              # I1 x I2 x I3 x F x N x M (src/UltrasoundSystem.m:3361)
    b0 = us.DAS(chd, interp="linear")
    assert tuple(b0.shape) == (30, 5, 1, 2, 1, 1)
    # (keep-dims runs the generic kernel -- fp32 delays like the reference; the summed call the tiled one: fp32-delay tolerance)
    assert rel_err(_np(b.sum((4, 5), keepdim=True)), _np(b0)) <= 1e-4
    assert rel_err(_np(b0[..., 1, :, :]), 2 * _np(b0[..., 0, :, :])) <= 1e-6
    chd_t = ChannelData(torch.from_numpy(np.swapaxes(xf, 1, 2)), case["t0"], case["fs"], order="TMN")
    assert rel_err(_np(us.DAS(chd_t, interp="linear")), _np(b0)) <= 1e-6


@pytest.mark.parametrize("keep", [(True, False), (False, True), (True, True), (False, False)])
def test_us_bfDAS_frames_layout_matches_DAS_and_oracle(keep):
    """bfDAS / bfDASLUT put the aperture dimensions BEHIND the frame dimensions like the reference
    (src/UltrasoundSystem.m:4663-4664) and like DAS (:3361): compared without any reshape, multi-frame, kept dimensions."""
    import torch
    from oracle import das_oracle as O
    from qups_amd import ChannelData, Scan, Sequence, Transducer, UltrasoundSystem
    keep_rx, keep_tx = keep
    case = make_case(seq="FSA", interp="linear", seed=23, N=5, I1=24, I2=4)
    N = M = 5
    xdc = Transducer(case["Pr"], np.stack([0 * case["Pr"][0], 0 * case["Pr"][0], 1 + 0 * case["Pr"][0]]))
    us = UltrasoundSystem(xdc, Sequence("FSA", c0=case["c"]), Scan(case["Pi"]))
    xf = np.stack([case["x"], 2j * case["x"], -0.5 * case["x"]], axis=3)      # T x N x M x F, F = 3
    chd = ChannelData(torch.from_numpy(xf), case["t0"], case["fs"])
    b_lut = us.bfDAS(chd, interp="linear", keep_rx=keep_rx, keep_tx=keep_tx)
    b_das = us.DAS(chd, interp="linear", keep_rx=keep_rx, keep_tx=keep_tx)
    want = (24, 4, 1, 3, N if keep_rx else 1, M if keep_tx else 1)
    assert tuple(b_lut.shape) == want and tuple(b_das.shape) == want
    fun = {(False, False): "DAS", (True, False): "SYN", (False, True): "MUL", (True, True): "BF"}[keep]
    ref1 = O.das_spec(fun, case["Pi"], case["Pr"], case["Pv"], case["Nv"], case["x"], case["t0"], case["fs"], cinv_f32(case["c"]),
                      VS=case["VS"], DV=case["DV"], interp="linear")          # I1 x I2 x I3 x [N] x [M]
    ref = np.stack([ref1, 2j * ref1, -0.5 * ref1], axis=3)                    # I1 x I2 x I3 x F x [N] x [M]
    assert ref.shape == want
    assert rel_err(_np(b_das), ref) <= 1e-4
    assert rel_err(_np(b_lut), ref) <= 5e-4                                   # fp32 delay tables


@pytest.mark.parametrize("shape", ["row", "column", "dim4", "scalar-array"])
def test_channeldata_arrays_in_DAS_and_bfDAS(shape):
    """ND-arrays of ChannelData (reference src/UltrasoundSystem.m:3301-3305,3325,3368 and :4629,4670-4672): every element is beamformed on
    its own and the images are concatenated along the array's one non-scalar dimension -- the reference's ``cat(chddim, bi{:})``: a row
    ``[chd1, chd2, chd3]`` concatenates along the image's second dimension, a 1 x 1 x 1 x K array (``splice(chd, 4, 1)``) along the frame
    dimension."""
    import torch
    from qups_amd import ChannelData, Scan, Sequence, Transducer, UltrasoundSystem
    case = make_case(seq="FSA", interp="linear", seed=31, N=5, I1=20, I2=4)
    xdc = Transducer(case["Pr"], np.stack([0 * case["Pr"][0], 0 * case["Pr"][0], 1 + 0 * case["Pr"][0]]))
    us = UltrasoundSystem(xdc, Sequence("FSA", c0=case["c"]), Scan(case["Pi"]))
    scal = [1.0, 2j, -0.5]
    one = [ChannelData(torch.from_numpy(case["x"] * s), case["t0"], case["fs"]) for s in scal]
    if shape == "row":
        arr, dim = one, 1                                                     # a list is MATLAB's [a, b, c]: 1 x 3
    elif shape == "column":
        arr, dim = np.array(one, dtype=object).reshape(3, 1), 0
    elif shape == "dim4":
        arr, dim = np.array(one, dtype=object).reshape(1, 1, 1, 3), 3
    else:
        arr, dim = np.array(one[:1], dtype=object).reshape(1, 1), None
    single = [us.DAS(c, interp="linear") for c in one]                        # I1 x I2 x I3 each
    for f in (us.DAS, us.bfDAS):
        b = f(arr, interp="linear")
        refs = [f(c, interp="linear") for c in one]
        if dim is None:
            assert tuple(b.shape) == tuple(refs[0].shape) and torch.equal(b, refs[0])
            continue
        nd = max(refs[0].ndim, dim + 1)
        want = torch.cat([r.reshape(tuple(r.shape) + (1,) * (nd - r.ndim)) for r in refs], dim=dim)
        assert tuple(b.shape) == tuple(want.shape) and torch.equal(b, want)
        assert b.shape[dim] == 3 * want.shape[dim] // 3 and b.shape[dim] == (3 if dim >= 3 else 3 * single[0].shape[dim])
    b, plan = us.DAS(arr, interp="linear", return_plan=True)
    assert plan is not None and not plan.closed
    plan.close()
    with pytest.raises(Exception, match="up to one non-scalar dimension"):
        us.DAS(np.array(one + one[:1], dtype=object).reshape(2, 2), interp="linear")


@pytest.mark.parametrize("keep_tx", [False, True])
def test_bfDASLUT_transmit_blocks(keep_tx):
    """bfDASLUT's transmit blocking (reference src/UltrasoundSystem.m:4573,4641-4655: `bsize`, 1 GB heuristic): the transmits are beamformed
    block by block, weights with a transmit dimension are indexed per block, blocks are summed or -- keep_tx -- concatenated; the result does
    not depend on the block size, per-transmit start times included"""
    import torch
    from oracle import das_oracle as O
    from qups_amd import ChannelData, Scan, Sequence, Transducer, UltrasoundSystem
    case = make_case(seq="PW", interp="linear", seed=29, N=6, M=7, I1=30, I2=5)
    N, M = 6, 7
    xdc = Transducer(case["Pr"], np.stack([0 * case["Pr"][0], 0 * case["Pr"][0], 1 + 0 * case["Pr"][0]]))
    us = UltrasoundSystem(xdc, Sequence("PW", focus=case["Nv"], c0=case["c"]), Scan(case["Pi"]))
    t0 = (case["t0"] + np.arange(M) / case["fs"]).reshape(1, 1, M)
    chd = ChannelData(torch.from_numpy(case["x"]), t0, case["fs"])
    rng = np.random.default_rng(3)
    a_nm = rng.uniform(0.2, 1.0, (1, 1, 1, N, M)).astype(np.float32)            # receiver x transmit
    a_pm = rng.uniform(0.2, 1.0, (30, 5, 1, 1, M)).astype(np.float32)           # pixel x transmit
    a_n = rng.uniform(0.2, 1.0, (1, 1, 1, N)).astype(np.float32)
    outs = [us.bfDAS(chd, a_nm, a_pm, a_n, interp="linear", keep_tx=keep_tx, bsize=bs) for bs in (None, 1, 3, M)]
    fun = "MUL" if keep_tx else "DAS"
    ref = O.das_spec(fun, case["Pi"], case["Pr"], case["Pv"], case["Nv"], case["x"], t0, case["fs"], cinv_f32(case["c"]),
                     VS=case["VS"], DV=case["DV"], interp="linear", apod=[a_nm.astype(np.float64), a_pm.astype(np.float64), a_n.reshape(1, 1, 1, N, 1).astype(np.float64)])
    for b in outs:
        assert tuple(b.shape) == (30, 5, 1, 1, M if keep_tx else 1)
        assert rel_err(_np(b).reshape(ref.shape), ref) <= 5e-4               # fp32 delay tables
        assert rel_err(_np(b), _np(outs[0])) <= 5e-5               # (other kernels / other products of the weight factors per block size)
    with pytest.raises(ValueError):
        us.bfDAS(chd, interp="linear", bsize=0)


def test_all_32_apodization_shapes_through_DAS_and_bfDAS():
    """reference test/USTest.m:333-337 (bfordgeneric): every broadcastable apodization shape -- each of I1, I2, I3, N, M full or
    singleton -- through UltrasoundSystem.DAS and bfDAS.  The reference asserts the output size (:296); here also the values."""
    import itertools
    import torch
    from oracle import das_oracle as O
    from qups_amd import ChannelData, Scan, Sequence, Transducer, UltrasoundSystem
    case = make_case(seq="PW", interp="linear", seed=27, N=5, M=3, I1=12, I2=4)
    xdc = Transducer(case["Pr"], np.stack([0 * case["Pr"][0], 0 * case["Pr"][0], 1 + 0 * case["Pr"][0]]))
    us = UltrasoundSystem(xdc, Sequence("PW", focus=case["Nv"], c0=case["c"]), Scan(case["Pi"]))
    chd = ChannelData(torch.from_numpy(case["x"]), case["t0"], case["fs"])
    full = (12, 4, 1, 5, 3)
    rng = np.random.default_rng(8)
    shapes = sorted({tuple(1 if m else f for m, f in zip(mask, full)) for mask in itertools.product((0, 1), repeat=5)})
    assert len(shapes) == 16                                          # I3 == 1: 2^4 distinct shapes of the 32 masks
    for shp in shapes:
        a = (0.5 + rng.random(shp)).astype(np.float32)
        ref = O.das_spec("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], case["x"], case["t0"], case["fs"], cinv_f32(case["c"]),
                         VS=case["VS"], DV=case["DV"], interp="linear", apod=[a])
        b1, b2 = us.DAS(chd, a, interp="linear"), us.bfDAS(chd, a, interp="linear")
        assert tuple(b1.shape[:3]) == (12, 4, 1) and tuple(b2.shape[:3]) == (12, 4, 1)       # test/USTest.m:296
        assert rel_err(_np(b1).reshape(ref.shape), ref) <= 1e-4, shp
        assert rel_err(_np(b2).reshape(ref.shape), ref) <= 5e-4, shp


@pytest.mark.parametrize("prec", ["single", "double", "halfT"])
def test_greens_then_DAS_has_no_nan_and_is_not_all_zero(prec):
    """reference test/ParTest.m:185-209 (greens_das_dev -> logTestCheck): simulate, beamform; no NaN, not all zero"""
    import torch
    from qups_amd import Scan, Sequence, Transducer, UltrasoundSystem
    fc, c0 = 5e6, 1500.0
    fs = 4 * fc
    xdc = Transducer.linear(16, 0.3e-3, fc)
    us = UltrasoundSystem(xdc, Sequence("FSA", c0=c0), Scan.cartesian(np.linspace(-2e-3, 2e-3, 21), np.linspace(8e-3, 12e-3, 33)), fs=fs)
    t = np.arange(-2.0 / fc, 2.0 / fc, 1 / (4 * fs))
    wv = np.exp(-(t * fc * 1.2) ** 2) * np.exp(2j * np.pi * fc * t)
    chd = us.greens(np.array([[0.5e-3], [0.0], [10e-3]]), [1.0], wv, t[0], 4 * fs, R0=c0 / fc)
    d = _np(chd.data)
    assert not np.isnan(d).any() and np.count_nonzero(d)
    b = us.DAS(chd, prec=prec)
    bn = torch.view_as_real(b).float().cpu().numpy() if prec == "halfT" else _np(b)
    assert not np.isnan(bn).any() and np.count_nonzero(bn)
