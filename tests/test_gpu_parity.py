"""GPU parity tests: the HIP path (through the C ABI of libqdas.so) against the float64 oracle on
identical float32 inputs.  Stated tolerances:

    fp32 data:  max|b - b_ref| / max|b_ref| <= 1e-4      (generic kernel: the reference's own fp32 time
                                                          arithmetic; tiled kernel is ~10x tighter)
    fp16 data:  <= 2e-3 against the oracle fed with the SAME half-rounded data (fp32 accumulation)
    fp64 data:  <= 1e-10
"""
import numpy as np
import pytest

from qups_amd import geometry as G
from tests.cases import cinv_f32, make_case, rel_err

pytestmark = pytest.mark.gpu

TOL32 = 1e-4


def tol_for(interp, kernel):
    """'nearest' is discontinuous in tau: an fp32 rounding of tau*fs across a half-integer picks the
    neighbouring sample for that one (n, m) term, so only a statistical bound can be stated."""
    if interp == "nearest":
        return 1e-2 if kernel == 1 else 2e-3
    return TOL32 if kernel == 1 else 2e-5


def _torch():
    import torch
    return torch


def run_das(case, fun="DAS", kernel=0, prec="single", apod=(), fmod=0.0, tpose=False, c=None, x=None, t0=None):
    torch = _torch()
    from qups_amd import das_spec
    x = case["x"] if x is None else x
    xs = np.swapaxes(x, 1, 2) if tpose else x
    opts = list(case["opt"]) + ["interp", case["interp"], "input-precision", prec, "modulation", fmod, "transpose", tpose]
    for a in apod:
        opts += ["apod", a]
    y, plan = das_spec(fun, case["Pi"], case["Pr"], case["Pv"], case["Nv"], torch.from_numpy(np.ascontiguousarray(xs)),
                       case["t0"] if t0 is None else t0, case["fs"], case["c"] if c is None else c, *opts,
                       return_plan=True, kernel=kernel)
    torch.cuda.synchronize()
    out = y.to(torch.complex128 if prec == "double" else torch.complex64).cpu().numpy()
    return out, plan


def run_oracle(case, fun="DAS", apod=(), fmod=0.0, c=None, x=None, t0=None, interp=None):
    from oracle import das_oracle as O
    c = case["c"] if c is None else c
    return O.das_spec(fun, case["Pi"], case["Pr"], case["Pv"], case["Nv"], case["x"] if x is None else x,
                      case["t0"] if t0 is None else t0, case["fs"], cinv_f32(c) if np.isscalar(c) else c,
                      VS=case["VS"], DV=case["DV"], interp=interp or case["interp"], apod=apod, fmod=fmod)


def test_extension_loaded_and_device():
    from qups_amd import _lib
    info = _lib.device_info()
    assert "gfx950" in info["name"], info
    assert info["cu_count"] >= 200


@pytest.mark.parametrize("interp", ["nearest", "linear", "cubic", "lanczos3", "cubic_dev"])
@pytest.mark.parametrize("seq", ["FSA", "PW", "FC", "DV"])
def test_das_generic_matches_oracle(seq, interp):
    case = make_case(seq=seq, interp=interp, seed=1)
    ref = run_oracle(case)
    out, plan = run_das(case, kernel=1)
    assert plan.kernel == "generic"
    assert np.abs(ref).max() > 0
    assert rel_err(out, ref) <= tol_for(interp, 1)


@pytest.mark.parametrize("interp", ["nearest", "linear", "cubic", "lanczos3", "cubic_dev"])
@pytest.mark.parametrize("seq", ["FSA", "PW", "FC", "DV"])
def test_das_tiled_matches_oracle(seq, interp):
    case = make_case(seq=seq, interp=interp, seed=2, I1=150, I2=19)   # ragged tile edges in both axes
    ref = run_oracle(case)
    out, plan = run_das(case, kernel=2)
    assert plan.kernel == "tiled"
    # 'FC': the focused-wave delay flips sign at the focal depth (copysign, src/bf.cu:107), so the tile
    # that contains the focus has a huge delay spread and is legitimately redone by the generic kernel
    if seq == "FC":
        assert plan.fallback_tiles() <= 4
        assert rel_err(out, ref) <= tol_for(interp, 1)
    else:
        assert plan.fallback_tiles() == 0
        assert rel_err(out, ref) <= tol_for(interp, 2)


@pytest.mark.parametrize("interp,prec", [("cubic", "single"), ("lanczos3", "single"), ("linear", "halfT"), ("nearest", "single")])
def test_focused_transmits_with_the_foci_inside_the_image(interp, prec, monkeypatch):
    """the delay of a focused transmit flips sign at the plane through its focus (src/bf.cu:106-108): a tile that the plane crosses has no
    window that fits.  The plan lists every transmit twice -- once per side of its plane, the pixels of the other side weighted 0 -- and
    the image runs fused without fallback tiles (das_tile_impl.h / tile_params.h kindS == 3); weights and per-transmit t0 included"""
    rng = np.random.default_rng(31)
    case = make_case(seq="FC", interp=interp, seed=35, N=24, M=10, I1=190, I2=40, zlim=(3e-3, 17e-3), xspan=8e-3)
    x = case["x"]
    q = (lambda a: a.astype(np.float16).astype(np.float64)) if prec == "halfT" else f32r
    if prec == "halfT":
        x = x.real.astype(np.float16).astype(np.float64) + 1j * x.imag.astype(np.float16).astype(np.float64)
    wn = q(rng.uniform(0.3, 1, (1, 1, 1, 24, 1)))
    wm = q(rng.uniform(0.3, 1, (1, 1, 1, 1, 10)))
    t0 = (case["t0"] + np.float32(1.0 / case["fs"]) * rng.integers(-2, 3, (1, 1, 10))).astype(np.float32).astype(np.float64)
    ref = run_oracle(case, apod=(wn, wm), x=x, t0=t0)
    out, plan = run_das(case, kernel=2, prec=prec, apod=(wn, wm), t0=t0)
    assert plan.kernel == "tiled" and "roles swapped" in plan.kernel_name() and plan.fallback_tiles() == 0, (plan.kernel_name(), plan.fallback_tiles())
    # at the planes themselves the fp32 geometry of the kernel and the float64 oracle may disagree on the sign of a dot product that is ~0
    e = np.abs(out - ref).reshape(-1) / np.abs(ref).max()
    tol = {"single": 1e-2 if interp == "nearest" else 1e-4, "halfT": 3e-3}[prec]
    assert np.median(e) <= tol / 10 and (e > tol).mean() <= 0.01, (np.median(e), (e > tol).mean(), e.max())
    monkeypatch.setenv("QDAS_NO_SIDE_SPLIT", "1")
    monkeypatch.setenv("QDAS_NO_WIDE", "1")                               # (nor the 384-sample windows that would swallow this small case)
    out2, plan2 = run_das(case, kernel=2, prec=prec, apod=(wn, wm), t0=t0)
    if prec == "single":
        assert plan2.fallback_tiles() > 0                             # (what the plan was before: those tiles on the generic kernel; fp16 windows are twice as long)
    e2 = np.abs(out2 - out).reshape(-1) / np.abs(ref).max()
    assert (e2 > tol).mean() <= 0.01


@pytest.mark.parametrize("interp,prec", [("cubic", "single"), ("linear", "halfT")])
@pytest.mark.parametrize("rule", ["multiline", "roi", "multiline-x-acceptance"])
def test_focused_transmits_inside_the_image_with_pixel_weights(rule, interp, prec, monkeypatch):
    """the same two-sided plan with a multiline transmit apodization in the reference's shape (1 x I2 x 1 x 1 x M, src/UltrasoundSystem.m:5071)
    or a pixel-only gain: the stage weight is the side rule times the array (tile_params.h gen_kind 6); no fallback tiles"""
    from qups_amd import apodization as A
    rng = np.random.default_rng(32)
    case = make_case(seq="FC", interp=interp, seed=36, N=24, M=10, I1=190, I2=40, zlim=(3e-3, 17e-3), xspan=8e-3)
    x = case["x"]
    q = (lambda a: a.astype(np.float16).astype(np.float64)) if prec == "halfT" else f32r
    if prec == "halfT":
        x = x.real.astype(np.float16).astype(np.float64) + 1j * x.imag.astype(np.float16).astype(np.float64)
    extra = ()
    if rule.startswith("multiline"):
        a = q(A.ap_multiline(np.linspace(-4e-3, 4e-3, 40), np.asarray(case["Pv"])[0]))
        assert a.shape == (1, 40, 1, 1, 10)
        if rule == "multiline-x-acceptance":          # + a receive-side mask: per-pair pixel weights (fp32 data only: generic kernel otherwise)
            if prec != "single":
                pytest.skip("two-sided pixel weights run fused for fp32 data")
            nrm = np.asarray(G.linear_array(24, 0.3e-3)[1], np.float32).astype(np.float64)
            extra = (q(A.ap_acceptance_angle(case["Pi"], case["Pr"], nrm, 35.0)),)
    else:
        a = q(rng.uniform(0.5, 2.0, (190, 40, 1, 1, 1)))
        a[:, 33:] = 0.0
    wn = q(rng.uniform(0.3, 1, (1, 1, 1, 24, 1)))
    ref = run_oracle(case, apod=(a, wn) + extra, x=x)
    out, plan = run_das(case, kernel=2, prec=prec, apod=(a, wn) + extra)
    assert plan.kernel == "tiled" and plan.fallback_tiles() == 0, (plan.kernel_name(), plan.fallback_tiles())
    e = np.abs(out - ref).reshape(-1) / np.abs(ref).max()
    tol = {"single": 1e-4, "halfT": 3e-3}[prec]
    assert np.median(e) <= tol / 10 and (e > tol).mean() <= 0.01, (np.median(e), (e > tol).mean(), e.max())
    monkeypatch.setenv("QDAS_NO_SIDE_SPLIT", "1")
    monkeypatch.setenv("QDAS_NO_WIDE", "1")
    out2, plan2 = run_das(case, kernel=2, prec=prec, apod=(a, wn) + extra)
    if prec == "single" and not extra:
        assert plan2.fallback_tiles() > 0
    assert (np.abs(out2 - out).reshape(-1) / np.abs(ref).max() > tol).mean() <= 0.01


def test_tiled_and_generic_agree_on_noise():
    """white-noise data (what the reference's own benchmark feeds, test/ParTest.m:254-257)"""
    case = make_case(seq="FSA", interp="lanczos3", seed=3, data="noise", I1=130, I2=9)
    ref = run_oracle(case)
    out_t, _ = run_das(case, kernel=2)
    out_g, _ = run_das(case, kernel=1)
    assert rel_err(out_t, ref) <= 2e-5
    assert rel_err(out_g, ref) <= 3e-4        # fp32 time arithmetic on white noise


@pytest.mark.parametrize("fun", ["SYN", "MUL", "BF"])
def test_keep_modes(fun):
    case = make_case(seq="PW", interp="linear", seed=4, N=8, M=5, I1=40, I2=6)
    ref = run_oracle(case, fun=fun)
    out, plan = run_das(case, fun=fun)
    assert plan.kernel == "tiled"        # fp32 data: 'SYN' / 'MUL' planes are accumulated, 'BF' planes stored, by the fused kernel
    assert out.shape == ref.shape
    assert rel_err(out, ref) <= TOL32
    gen, _ = run_das(case, fun=fun, kernel=1)
    assert rel_err(gen, ref) <= TOL32
    # identity: summing the kept dimensions reproduces 'DAS' (kern/das_spec.m:263-269)
    das, _ = run_das(case, fun="DAS", kernel=1)
    assert rel_err(gen.sum(axis=(3, 4), keepdims=True), das) <= 1e-5        # same kernel: summation order only
    assert rel_err(out.sum(axis=(3, 4), keepdims=True), das) <= TOL32


def test_bf_transposed_output_order():
    case = make_case(seq="PW", interp="linear", seed=5, N=6, M=4, I1=33, I2=3)
    ref = run_oracle(case, fun="BF")                 # I x N x M
    out, _ = run_das(case, fun="BF", tpose=True)     # reference stores plane nm = m + n*M  (src/bf.cu:100,135)
    assert out.shape[3:5] == (case["M"], case["N"])
    assert rel_err(np.swapaxes(out, 3, 4), ref) <= TOL32


@pytest.mark.parametrize("kernel", [1, 2])
def test_transpose_and_per_tx_t0(kernel):
    case = make_case(seq="FSA", interp="cubic", seed=6, N=12)
    t0 = (case["t0"] + np.linspace(0, 3, case["M"]) / case["fs"]).astype(np.float32).astype(np.float64)
    x = np.stack([np.roll(case["x"][:, :, m], 0, axis=0) for m in range(case["M"])], axis=2)
    ref = run_oracle(case, x=x, t0=t0)
    out, plan = run_das(case, x=x, t0=t0, tpose=True, kernel=kernel)
    assert rel_err(out, ref) <= (TOL32 if kernel == 1 else 2e-5)


@pytest.mark.parametrize("kernel", [1, 2])
def test_fmod_device_semantics(kernel):
    case = make_case(seq="PW", interp="cubic", seed=7)
    fm = float(np.float32(case["fc"]))
    ref = run_oracle(case, fmod=fm)
    out, _ = run_das(case, fmod=fm, kernel=kernel)
    assert rel_err(out, ref) <= 2e-4     # phase 2*pi*fmod*tau with tau ~ 1e-5 s in fp32


def test_apod_broadcast_shapes_generic():
    """every singleton pattern of I1 x I2 x I3 x N x M (reference test/USTest.m:333-337)"""
    case = make_case(seq="PW", interp="linear", seed=8, N=5, M=3, I1=20, I2=4)
    rng = np.random.default_rng(0)
    full = (20, 4, 1, 5, 3)
    for mask in range(32):
        shp = tuple(full[k] if (mask >> k) & 1 else 1 for k in range(5))
        a = f32r(rng.uniform(0.2, 1.0, shp))
        ref = run_oracle(case, apod=(a,))
        out, _ = run_das(case, apod=(a,))
        assert rel_err(out, ref) <= TOL32, shp


def f32r(a):
    return np.asarray(a, np.float32).astype(np.float64)


@pytest.mark.parametrize("kernel,name", [(1, "generic"), (0, "tiled")])
def test_apod_stack_complex_and_zero_skip(kernel, name):
    case = make_case(seq="FSA", interp="cubic", seed=9, N=8, I1=64, I2=8)
    rng = np.random.default_rng(1)
    a1 = f32r(rng.uniform(0, 1, (64, 8, 1, 8, 1)) > 0.4).astype(np.float64)          # sparse mask (zeros short-circuit)
    a2 = (f32r(rng.uniform(0, 1, (1, 1, 1, 1, 8))) + 1j * f32r(rng.uniform(0, 1, (1, 1, 1, 1, 8))))
    ref = run_oracle(case, apod=(a1, a2))
    out, plan = run_das(case, apod=(a1, a2), kernel=kernel)
    assert plan.kernel == name          # pixel x receiver mask -> in-kernel weights, transmit vector -> folded table
    assert rel_err(out, ref) <= TOL32


@pytest.mark.parametrize("seq,kernel", [("PW", 2), ("FSA", 2), ("PW", 1)])
def test_zero_weights_never_sample_their_trace(seq, kernel):
    """device semantics (src/bf.cu:120-126): a pair whose weight is zero is skipped, so non-finite samples of a dead channel or an
    unused transmit stay out of the image -- also when the zero shares a packed transmit pair with a non-zero weight, and in the
    reciprocal mode (FSA: both w[n,m] and w[m,n])"""
    rng = np.random.default_rng(21)
    case = make_case(seq=seq, interp="cubic", seed=27, N=16, M=16, I1=100, I2=12)
    x = case["x"].copy()
    wn = f32r(rng.uniform(0.3, 1, (1, 1, 1, 16, 1)))
    wm = f32r(rng.uniform(0.3, 1, (1, 1, 1, 1, 16)))
    wn[..., 5, :] = 0.0                                   # dead receive channel 5, unused transmits 2 and 11 (isolated zeros)
    wm[..., [2, 11]] = 0.0
    xz = x.copy()
    xz[:, 5, :] = 0
    xz[:, :, [2, 11]] = 0
    x[:, 5, :] = np.nan
    x[:, :, 2] = np.inf
    x[:, :, 11] = np.nan
    ref = run_oracle(case, apod=(wn, wm), x=xz)
    out, plan = run_das(case, kernel=kernel, apod=(wn, wm), x=x)
    assert np.isfinite(out).all()
    assert rel_err(out, ref) <= (3e-5 if kernel == 2 else TOL32)
    if kernel == 2:
        assert plan.kernel == "tiled" and plan.reciprocal == (seq == "FSA")


def test_tiled_with_trace_weights():
    case = make_case(seq="FSA", interp="lanczos3", seed=10, N=10, I1=100, I2=8)
    rng = np.random.default_rng(2)
    an = f32r(np.hanning(12)[1:-1]).reshape(1, 1, 1, 10, 1)
    am = (f32r(rng.uniform(0, 1, (1, 1, 1, 1, 10))) > 0.3) * (1 + 0.5j)
    ref = run_oracle(case, apod=(an, am))
    out, plan = run_das(case, apod=(an, am), kernel=2)
    assert plan.kernel == "tiled"
    assert rel_err(out, ref) <= 2e-5


def test_nd_sound_speed():
    case = make_case(seq="PW", interp="linear", seed=11, N=6, M=4, I1=30, I2=5)
    rng = np.random.default_rng(3)
    cmap = f32r(1.0 / f32r(1.0 / rng.uniform(1480, 1600, (30, 5, 1))))
    from oracle import das_oracle as O
    ref = O.das_spec("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], case["x"], case["t0"], case["fs"],
                     1.0 / f32r(1.0 / cmap), VS=case["VS"], DV=case["DV"], interp="linear")
    out, plan = run_das(case, c=cmap, kernel=1)
    assert plan.kernel == "generic"
    assert rel_err(out, ref) <= TOL32
    # the tiled kernel takes per-pixel maps too (the delay stays separable); white-noise speeds give tiles whose delay spread
    # exceeds the LDS window: those are redone by the generic kernel -- same image
    out, plan = run_das(case, c=cmap)
    assert plan.kernel == "tiled"
    assert rel_err(out, ref) <= TOL32


@pytest.mark.parametrize("seq,prec", [("FSA", "single"), ("PW", "single"), ("DV", "halfT")])
def test_sound_speed_map_in_the_tiled_kernel(seq, prec):
    """a smooth I1 x I2 sound-speed map (aberration correction): fused kernel, no fallback, reciprocal mode included"""
    case = make_case(seq=seq, interp="cubic", seed=17, N=16, I1=140, I2=24, zlim=(4e-3, 18e-3), xspan=4e-3)
    zz, xx = np.meshgrid(np.linspace(0, 1, 140), np.linspace(0, 1, 24), indexing="ij")
    cmap = f32r(1.0 / f32r(1.0 / (1540.0 + 25.0 * np.sin(2.1 * zz + 0.4) * np.cos(1.7 * xx))))[:, :, None]
    xq = case["x"]
    if prec == "halfT":
        xq = xq.real.astype(np.float16).astype(np.float64) + 1j * xq.imag.astype(np.float16).astype(np.float64)
    from oracle import das_oracle as O
    ref = O.das_spec("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], xq, case["t0"], case["fs"],
                     1.0 / f32r(1.0 / cmap), VS=case["VS"], DV=case["DV"], interp="cubic")
    out, plan = run_das(case, c=cmap, prec=prec, kernel=2)
    assert plan.kernel == "tiled" and plan.fallback_tiles() == 0
    assert rel_err(out, ref) <= (2e-3 if prec == "halfT" else 2e-5)
    gen, _ = run_das(case, c=cmap, prec=prec, kernel=1)
    assert rel_err(gen, ref) <= (2e-3 if prec == "halfT" else TOL32)


def test_double_precision():
    case = make_case(seq="FC", interp="cubic", seed=12)
    from oracle import das_oracle as O
    ref = O.das_spec("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], case["x"], case["t0"], case["fs"], case["c"],
                     VS=case["VS"], DV=case["DV"], interp="cubic")
    out, plan = run_das(case, prec="double")
    assert rel_err(out, ref) <= 1e-10


def _oracle64(case, apod=(), x=None, t0=None, fmod=0.0):
    from oracle import das_oracle as O
    return O.das_spec("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], case["x"] if x is None else x, case["t0"] if t0 is None else t0,
                      case["fs"], case["c"], VS=case["VS"], DV=case["DV"], interp=case["interp"], apod=apod, fmod=fmod)


@pytest.mark.parametrize("interp", ["nearest", "linear", "cubic", "lanczos3", "cubic_dev"])
@pytest.mark.parametrize("seq", ["FSA", "PW", "FC", "DV"])
def test_double_precision_in_the_tiled_kernel(seq, interp):
    """fp64 data on the fused kernel (launch configuration 13): geometry, delays, weights and sums in double -- the bar of this
    precision is 1e-10 of the image maximum, as for the generic kernel (reference twin: `DAS`, src/bf.cu:144-151)"""
    case = make_case(seq=seq, interp=interp, seed=3, I1=150, I2=19)          # ragged tile edges in both axes
    ref = _oracle64(case)
    out, plan = run_das(case, kernel=2, prec="double")
    assert plan.kernel == "tiled" and ",f64" in plan.kernel_name(), plan.kernel_name()
    if seq != "FC":
        assert plan.fallback_tiles() == 0
    assert rel_err(out, ref) <= 1e-10
    gen, _ = run_das(case, kernel=1, prec="double")
    assert rel_err(out, gen) <= 1e-10


@pytest.mark.parametrize("interp", ["nearest", "linear", "cubic", "lanczos3", "cubic_dev"])
@pytest.mark.parametrize("seq", ["FSA", "PW", "DV"])
def test_double_precision_remodulation_in_the_tiled_kernel(seq, interp):
    """fp64 baseband data on the fused kernel: the remodulation phasor exp(2j pi fmod tau) of every sample (src/bf.cu:117, cospi / sinpi in
    double) -- phase in fp64 from the window bases, quarter-turn reduction + polynomial -- to the 1e-10 bar of this precision"""
    case = make_case(seq=seq, interp=interp, seed=5, I1=150, I2=19, M=9 if seq != "FSA" else None)        # (odd transmit count: tail block)
    fmod = 3.1e6
    ref = _oracle64(case, fmod=fmod)
    out, plan = run_das(case, kernel=2, prec="double", fmod=fmod)
    assert plan.kernel == "tiled" and ",f64" in plan.kernel_name() and ",fmod" in plan.kernel_name(), plan.kernel_name()
    assert plan.fallback_tiles() == 0
    assert rel_err(out, ref) <= 1e-10
    gen, _ = run_das(case, kernel=1, prec="double", fmod=fmod)
    assert rel_err(out, gen) <= 1e-10


@pytest.mark.parametrize("interp,fmod,wtab", [("cubic", 0.0, False), ("lanczos3", 3.1e6, True), ("linear", -2.0e6, False), ("nearest", 0.0, True)])
def test_double_precision_hiprtc_build(interp, fmod, wtab, tmp_path, monkeypatch):
    """fp64 plans take a plan-specialised (hiprtc) build as well: same numbers as the prebuilt kernel to re-association, 1e-10 against the oracle"""
    torch = _torch()
    from qups_amd import DasPlan, build_problem, parse_options
    monkeypatch.setenv("QDAS_CACHE_DIR", str(tmp_path))
    rng = np.random.default_rng(7)
    case = make_case(seq="PW", interp=interp, seed=23, I1=150, I2=21, N=24, M=19)
    N, M = case["N"], case["M"]
    ap = ()
    va = list(case["opt"]) + ["interp", interp, "input-precision", "double", "modulation", fmod]
    if wtab:
        ap = (rng.uniform(0.2, 1, (1, 1, 1, N, 1)), rng.uniform(0.1, 1, (1, 1, 1, 1, M)) * (1 + 0.5j))
        for a in ap:
            va += ["apod", a]
    xt = torch.from_numpy(case["x"].astype(np.complex128))
    prob = build_problem("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], tuple(xt.shape), case["t0"], case["fs"], case["c"], parse_options(xt, va))
    ys = []
    for jit in (False, True):
        with DasPlan(prob, kernel=2, jit=jit) as plan:
            assert ("[jit " in plan.kernel_name()) == jit and ",f64" in plan.kernel_name(), plan.kernel_name()
            ys.append(plan.feval(xt).cpu().numpy().reshape(-1))
    ref = _oracle64(case, apod=ap, fmod=fmod).reshape(-1, order="F")
    assert rel_err(ys[1], ref) <= 1e-10 and rel_err(ys[0], ref) <= 1e-10
    assert rel_err(ys[1], ys[0]) <= 1e-12


@pytest.mark.parametrize("kind", ["real", "complex", "pixel-only", "mask"])
@pytest.mark.parametrize("interp", ["linear", "lanczos3"])
def test_double_precision_pixel_by_receiver_weights_in_the_tiled_kernel(kind, interp):
    """fp64 data with ONE pixel x receiver (or pixel-only) apodization array: the fused kernel multiplies every stage's sum by the stage
    element's weight in double (plain list of stages; weightless stages of a wave read no sample) -- 1e-10 against the oracle; with
    remodulation, a second pixel array or a pixel x transmit array the plan is the generic kernel's"""
    rng = np.random.default_rng(11)
    case = make_case(seq="PW", interp=interp, seed=31, I1=140, I2=22, N=20, M=11)
    I1, I2, N = 140, 22, case["N"]
    if kind == "real":
        a = rng.uniform(0.1, 1.0, (I1, I2, 1, N, 1))
    elif kind == "complex":
        a = rng.uniform(0.1, 1.0, (I1, I2, 1, N, 1)) * np.exp(1j * rng.uniform(0, 1, (I1, I2, 1, N, 1)))
    elif kind == "pixel-only":
        a = rng.uniform(0.0, 1.0, (I1, I2, 1)) * (rng.random((I1, I2, 1)) > 0.3)
    else:
        a = (rng.random((I1, I2, 1, N, 1)) > 0.5) * 1.0
        a[: I1 // 2, :, :, : N // 2] = 0.0                               # shallow pixels: half the aperture carries no weight at all
    ref = _oracle64(case, apod=(a,))
    out, plan = run_das(case, kernel=2, prec="double", apod=(a,))
    assert plan.kernel == "tiled" and ",f64" in plan.kernel_name() and plan.fallback_tiles() == 0, plan.kernel_name()
    assert rel_err(out, ref) <= 1e-10
    gen, _ = run_das(case, kernel=1, prec="double", apod=(a,))
    assert rel_err(out, gen) <= 1e-10
    if kind in ("real", "mask"):                                          # ... together with pixel-independent windows (an N x M table in double)
        wn = np.hanning(N + 2)[1:-1].reshape(1, 1, 1, N, 1)
        wm = (0.5 + 0.5 * np.hanning(case["M"] + 2)[1:-1]).reshape(1, 1, 1, 1, case["M"]) * (1 + 0.2j)
        o3, p3 = run_das(case, kernel=2, prec="double", apod=(wn, a, wm))
        assert p3.kernel == "tiled" and ",wtab" in p3.kernel_name()
        assert rel_err(o3, _oracle64(case, apod=(wn, a, wm))) <= 1e-10
    if kind == "real":                                                    # what stays on the generic kernel
        for kw in (dict(fmod=2.0e6), dict(apod=(a, a))):
            kw.setdefault("apod", (a,))
            o2, p2 = run_das(case, kernel=0, prec="double", **kw)
            assert p2.kernel == "generic"
            assert rel_err(o2, _oracle64(case, apod=kw["apod"], fmod=kw.get("fmod", 0.0))) <= 1e-10


def test_double_precision_remodulation_variants():
    """remodulation of fp64 data together with a weight table (one zero weight), with a record shorter than the path (checked loop) and
    with a negative modulation frequency"""
    rng = np.random.default_rng(6)
    case = make_case(seq="PW", interp="lanczos3", seed=22, I1=90, I2=23, N=24, M=19)
    N, M = case["N"], case["M"]
    wn = rng.uniform(0.2, 1, (1, 1, 1, N, 1))
    wm = rng.uniform(0, 1, (1, 1, 1, 1, M)) * (1 - 0.25j)
    wm[..., 5] = 0.0
    for fmod in (4.0e6, -2.5e6):
        out, plan = run_das(case, kernel=2, prec="double", apod=(wn, wm), fmod=fmod)
        assert plan.kernel == "tiled" and ",wtab" in plan.kernel_name() and ",fmod" in plan.kernel_name()
        assert rel_err(out, _oracle64(case, apod=(wn, wm), fmod=fmod)) <= 1e-10
    for interp in ("nearest", "cubic"):
        case = make_case(seq="FSA", interp=interp, seed=14, T=300, data="noise", zlim=(1e-3, 30e-3), I1=128, I2=8)
        ref = _oracle64(case, fmod=2.0e6)
        out, plan = run_das(case, kernel=2, prec="double", fmod=2.0e6)
        dead = np.abs(ref) == 0
        assert plan.kernel == "tiled" and dead.any() and np.all(out[dead] == 0), interp
        assert rel_err(out, ref) <= 1e-10, interp


@pytest.mark.parametrize("ks", [2, 3, 4])
def test_double_precision_aperture_split(ks, monkeypatch):
    """few tiles (a small image, a pixel slab of a multi-GPU job): several workgroups per tile, complex128 partial images, fixed-order sum"""
    monkeypatch.setenv("QDAS_KSPLIT", str(ks))
    case = make_case(seq="DV", interp="lanczos3", seed=8, N=24, I1=100, I2=21)
    out, plan = run_das(case, kernel=2, prec="double")
    assert plan.kernel == "tiled" and plan.aperture_split() == ks
    assert rel_err(out, _oracle64(case)) <= 1e-10


def test_double_precision_tiled_variants():
    """weights (pixel-independent: folded into one N x M complex128 table), transposed data with per-transmit t0, a pixel shard,
    several frames, a record shorter than the path (checked loop: exact zeros) and an odd transmit count (tail block)"""
    torch = _torch()
    from qups_amd import DasPlan, build_problem, parse_options
    from qups_amd.das_spec import _cast_data, _colmajor
    rng = np.random.default_rng(5)
    case = make_case(seq="PW", interp="lanczos3", seed=21, I1=90, I2=23, N=24, M=19)
    N, M = case["N"], case["M"]
    wn = rng.uniform(0.2, 1, (1, 1, 1, N, 1))
    wm = rng.uniform(0, 1, (1, 1, 1, 1, M)) * (1 + 0.5j)
    wm[..., 3] = 0.0
    ref = _oracle64(case, apod=(wn, wm))
    out, plan = run_das(case, kernel=2, prec="double", apod=(wn, wm))
    assert plan.kernel == "tiled" and ",wtab" in plan.kernel_name()
    assert rel_err(out, ref) <= 1e-10
    # transposed data + per-transmit t0
    t0 = (case["t0"] + (1.0 / case["fs"]) * rng.integers(-3, 4, (1, 1, M))).astype(np.float64)
    ref = _oracle64(case, t0=t0)
    out, plan = run_das(case, kernel=2, prec="double", tpose=True, t0=t0)
    assert plan.kernel == "tiled"
    assert rel_err(out, ref) <= 1e-10
    # a pixel shard and three frames through one plan
    xs = [case["x"]] + [(rng.standard_normal(case["x"].shape) + 1j * rng.standard_normal(case["x"].shape)) for _ in range(2)]
    xt = torch.from_numpy(np.stack(xs, axis=3))
    po = parse_options(xt, list(case["opt"]) + ["interp", "lanczos3", "input-precision", "double"])
    prob = build_problem("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], tuple(xt.shape), case["t0"], case["fs"], case["c"], po)
    I = 90 * 23
    sh = DasPlan(prob, kernel=2, i_begin=I // 3, i_count=I // 2)
    y = sh.execute_colmajor(_colmajor(_cast_data(xt, "double", sh.device)), 3).cpu().numpy()     # (F, 1, 1, count)
    for f in range(3):
        r = _oracle64(case, x=xs[f]).reshape(-1, order="F")[I // 3: I // 3 + I // 2]
        assert np.abs(y[f].reshape(-1) - r).max() / np.abs(r).max() <= 1e-10, f
    # record shorter than the path: zeros stay exact zeros
    for interp in ("nearest", "linear", "cubic", "lanczos3"):
        case = make_case(seq="FSA", interp=interp, seed=14, T=300, data="noise", zlim=(1e-3, 30e-3), I1=128, I2=8)
        ref = _oracle64(case)
        out, plan = run_das(case, kernel=2, prec="double")
        dead = np.abs(ref) == 0
        assert dead.any() and np.all(out[dead] == 0), interp
        assert rel_err(out, ref) <= 1e-10, interp
    # delay spread beyond the LDS window: those tiles are redone by the generic fp64 kernel
    case = make_case(seq="FSA", interp="linear", seed=15, I1=64, I2=4, zlim=(2e-3, 60e-3), data="noise", N=8)
    out, plan = run_das(case, kernel=0, prec="double")
    assert plan.kernel == "tiled" and plan.fallback_tiles() > 0
    assert rel_err(out, _oracle64(case)) <= 1e-10


@pytest.mark.parametrize("kernel", [1, 2])
def test_half_precision(kernel):
    case = make_case(seq="FSA", interp="cubic", seed=13, I1=96, I2=8)
    xh = case["x"]
    xh = (xh.real.astype(np.float16).astype(np.float64) + 1j * xh.imag.astype(np.float16).astype(np.float64))
    ref = run_oracle(case, x=xh)
    out, plan = run_das(case, prec="halfT", kernel=kernel)
    assert rel_err(out, ref) <= 2e-3


def test_edges_and_out_of_record():
    """record shorter than the path: samples before 0 / after T-1 are exactly zero"""
    for interp in ("nearest", "linear", "cubic", "lanczos3"):
        case = make_case(seq="FSA", interp=interp, seed=14, T=300, data="noise", zlim=(1e-3, 30e-3), I1=128, I2=8)
        ref = run_oracle(case)
        for kernel in (1, 2):
            out, plan = run_das(case, kernel=kernel)
            # pixels whose every contribution is out of the record must be exactly zero
            dead = np.abs(ref) == 0
            assert dead.any()
            assert np.all(out[dead] == 0), (interp, kernel)
            tol = (2e-2 if kernel == 1 else 5e-3) if interp == "nearest" else (3e-4 if kernel == 1 else 3e-5)
            assert rel_err(out, ref) <= tol, (interp, kernel)


def test_oversize_window_falls_back(monkeypatch):
    """pixel spacing so coarse (24 samples of delay per pixel) that even the shortest tile spans more than the 192-sample LDS window: those
    tiles are redone by the generic kernel -- unless the plan's second attempt, 384-sample windows (launch configuration 14), fits them"""
    case = make_case(seq="FSA", interp="linear", seed=15, I1=64, I2=4, zlim=(2e-3, 60e-3), data="noise", N=8)
    ref = run_oracle(case)
    out, plan = run_das(case, kernel=0)
    assert plan.kernel == "tiled" and plan.fallback_tiles() == 0 and "W=384" in plan.kernel_name(), plan.kernel_name()
    assert rel_err(out, ref) <= 3e-5
    monkeypatch.setenv("QDAS_NO_WIDE", "1")
    out, plan = run_das(case, kernel=0)
    assert plan.kernel == "tiled" and plan.tile_shape()[0] < 64             # the footprint with the smallest misfit fraction
    assert plan.fallback_tiles() > 0
    assert rel_err(out, ref) <= 3e-4
    case = make_case(seq="PW", interp="cubic", seed=16, I1=60, I2=6, zlim=(2e-3, 120e-3), data="noise", N=8, M=4)      # 49 samples per pixel: beyond both
    monkeypatch.delenv("QDAS_NO_WIDE")
    out, plan = run_das(case, kernel=0)
    assert plan.fallback_tiles() > 0 and rel_err(out, run_oracle(case)) <= 3e-4


def test_delays():
    from qups_amd import das_spec
    from oracle import das_oracle as O
    case = make_case(seq="FC", seed=16, N=5, M=3, I1=17, I2=4)
    tau = das_spec("delays", case["Pi"], case["Pr"], case["Pv"], case["Nv"], None, 0.0, None, case["c"],
                   "input-precision", "single").cpu().numpy()
    ref = O.das_spec("delays", case["Pi"], case["Pr"], case["Pv"], case["Nv"], None, 0, 1, cinv_f32(case["c"]), VS=True, DV=False)
    assert tau.shape == ref.shape
    assert np.abs(tau - ref).max() <= 1e-6 * np.abs(ref).max()


def test_frames_and_plan_reuse():
    torch = _torch()
    from qups_amd import das_spec
    case = make_case(seq="PW", interp="cubic", seed=17, N=8, M=4, I1=70, I2=8)
    rng = np.random.default_rng(5)
    xf = np.stack([case["x"] * (k + 1) * np.exp(1j * k) for k in range(3)], axis=3).astype(np.complex64)
    y, plan = das_spec("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], torch.from_numpy(xf), case["t0"], case["fs"],
                       case["c"], *case["opt"], "interp", "cubic", return_plan=True)
    y = y.cpu().numpy()
    assert y.shape == (70, 8, 1, 1, 1, 3)
    ref = run_oracle(case)
    for k in range(3):
        assert rel_err(y[..., k], ref * (k + 1) * np.exp(1j * k)) <= 5e-5
    y1 = plan.feval(torch.from_numpy(xf[..., 1])).cpu().numpy()     # k.feval(PRE_ARGS{:}, x{f}, POST_ARGS{:})
    assert rel_err(y1.reshape(8, 70).T, y[:, :, 0, 0, 0, 1]) == 0.0   # I is column-major: i = i1 + 70*i2


def test_pixel_shards_concatenate():
    """multi-GPU layout on one device: contiguous slabs of the linear pixel index"""
    torch = _torch()
    from qups_amd import build_problem, parse_options, DasPlan
    case = make_case(seq="FSA", interp="cubic", seed=18, I1=100, I2=10)
    x = torch.from_numpy(case["x"]).cuda()
    opts = parse_options(x, list(case["opt"]) + ["interp", "cubic"])
    prob = build_problem("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], x.shape, case["t0"], case["fs"], case["c"], opts)
    full = DasPlan(prob, mirror=False).feval(x).cpu().numpy().reshape(-1)       # (slabs run without the lateral-mirror mode: same summation order)
    I = prob.I
    for G in (2, 3, 7):
        parts = []
        for g in range(G):
            b, e = I * g // G, I * (g + 1) // G
            parts.append(DasPlan(prob, i_begin=b, i_count=e - b).feval(x).cpu().numpy().reshape(-1))
        assert np.array_equal(np.concatenate(parts), full), G


@pytest.mark.parametrize("seq,fun", [("FSA", "DAS"), ("PW", "DAS"), ("DV", "SYN")])
def test_transmit_slabs_sum_to_the_image(seq, fun):
    """the alternative multi-GPU layout on one device: partial images over transmit slabs add up to the image"""
    torch = _torch()
    from qups_amd import das_spec
    from qups_amd.dist import slice_transmits
    case = make_case(seq=seq, interp="cubic", seed=21, N=16, M=None if seq == "FSA" else 9, I1=96, I2=16)
    M = case["M"]
    am = np.random.default_rng(3).uniform(0.3, 1, (1, 1, 1, 1, M))
    x = torch.from_numpy(case["x"]).cuda()
    opts = list(case["opt"]) + ["interp", "cubic"]
    full = das_spec(fun, case["Pi"], case["Pr"], case["Pv"], case["Nv"], x, case["t0"], case["fs"], case["c"], *opts, "apod", am)
    for G in (2, 3):
        acc = torch.zeros_like(full)
        for g in range(G):
            Pv, Nv, t0, ap, b, c = slice_transmits(case["Pv"], case["Nv"], case["t0"], [am], M, g, G)
            acc += das_spec(fun, case["Pi"], case["Pr"], Pv, Nv, x[:, :, b:b + c].contiguous(), t0, case["fs"], case["c"], *opts, "apod", ap[0])
        assert rel_err(acc.cpu().numpy(), full.cpu().numpy()) <= 2e-5, G


def test_c_abi_one_shot_matches_plan():
    """qdas_DASf: the reference kernel's own argument list (src/bf.cu:153-158)"""
    import ctypes as C
    torch = _torch()
    from qups_amd import _lib, build_problem, parse_options, DasPlan
    case = make_case(seq="PW", interp="linear", seed=19, N=6, M=3, I1=40, I2=4)
    x = torch.from_numpy(case["x"]).cuda()
    opts = parse_options(x, list(case["opt"]) + ["interp", "linear"])
    p = build_problem("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], x.shape, case["t0"], case["fs"], case["c"], opts)
    ref = DasPlan(p, kernel=1).feval(x).reshape(-1)
    L = _lib.lib()
    dev = lambda a: torch.from_numpy(a).cuda()
    Pi, Pr, Pv, Nv, cinv = dev(p.Pi), dev(p.Pr), dev(p.Pv), dev(p.Nv), dev(p.cinv)
    xc = x.permute(2, 1, 0).contiguous()
    y = torch.zeros(p.I, dtype=torch.complex64, device="cuda")
    sz = _lib.Sizes(p.T, p.N, p.M, p.Isz[0], p.Isz[1], p.Isz[2], 0, p.flag, int(p.VS), int(p.DV), 1)
    acs = (C.c_uint64 * 6)(*[0] * 6)
    tv = (C.c_float * 2)(p.fs, 0.0)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(L.qdas_DASf(C.byref(sz), ptr(y), ptr(Pi), ptr(Pr), ptr(Pv), ptr(Nv), None, ptr(cinv), acs, ptr(xc),
                           C.cast(tv, C.c_void_p), None))
    torch.cuda.synchronize()
    assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) <= 2e-5


def test_error_paths():
    from qups_amd import das_spec, DasError, _lib
    case = make_case(seq="PW", seed=20, N=4, M=3, I1=8, I2=2)
    torch = _torch()
    x = torch.from_numpy(case["x"])
    args = (case["Pi"], case["Pr"], case["Pv"], case["Nv"], x, case["t0"], case["fs"], case["c"])
    with pytest.raises(DasError, match="Unrecognized option"):
        das_spec("DAS", *args, "bogus")
    with pytest.raises(DasError, match="Invalid beamformer"):
        das_spec("XYZ", *args)
    with pytest.raises(DasError) as ei:
        das_spec("DAS", *args, "interp", "spline")
    assert ei.value.identifier == "QUPS:das_spec:UnrecognizedInput"
    with pytest.raises(DasError, match="Apodization data size inconsistent with receiver"):
        das_spec("DAS", *args, "apod", np.ones((1, 1, 1, 5, 1)))
    with pytest.raises(_lib.QdasError, match="tiled kernel"):           # forcing the fused kernel on a case it cannot serve says why
        das_spec("BF", *args, "input-precision", "double", kernel=2)


def test_reciprocal_mode_matches_general_mode(monkeypatch):
    """FSA with Pv == Pr runs the reciprocal (SYM) tiled kernel: tau(n,m) == tau(m,n), weights shared by the direct and the
    mirror trace.  It must agree with the general tiled kernel (QDAS_NO_SYM=1) and with the oracle, also for transposed data,
    and must NOT engage when the geometry is not reciprocal (per-transmit t0, N not a multiple of the transmit block)."""
    for tpose in (False, True):
        case = make_case(seq="FSA", interp="lanczos3", seed=31, N=32, I1=140, I2=20, data="noise")
        ref = run_oracle(case)
        monkeypatch.delenv("QDAS_NO_SYM", raising=False)
        a, pa = run_das(case, kernel=2, tpose=tpose)
        monkeypatch.setenv("QDAS_NO_SYM", "1")
        b, pb = run_das(case, kernel=2, tpose=tpose)
        monkeypatch.delenv("QDAS_NO_SYM", raising=False)
        assert pa.fallback_tiles() == 0 and pb.fallback_tiles() == 0
        assert rel_err(a, ref) <= 2e-5 and rel_err(b, ref) <= 2e-5
        assert rel_err(a, b) <= 1e-5
    # edge of the record + reciprocal mode (checked loop)
    case = make_case(seq="FSA", interp="cubic", seed=32, N=16, T=300, data="noise", zlim=(1e-3, 30e-3), I1=128, I2=16)
    ref = run_oracle(case)
    out, _ = run_das(case, kernel=2)
    assert np.all(out[np.abs(ref) == 0] == 0) and rel_err(out, ref) <= 3e-5
    # fmod in reciprocal mode
    case = make_case(seq="FSA", interp="cubic", seed=33, N=16, I1=100, I2=16)
    fm = float(np.float32(case["fc"]))
    assert rel_err(run_das(case, kernel=2, fmod=fm)[0], run_oracle(case, fmod=fm)) <= 2e-4
    # not reciprocal: per-transmit t0 -> general kernel path, still correct
    case = make_case(seq="FSA", interp="linear", seed=34, N=16, I1=70, I2=16)
    t0 = (case["t0"] + np.arange(16) / case["fs"]).astype(np.float32).astype(np.float64)
    # (this grid is coarse enough that its first tile overflows the LDS window and is redone by the generic kernel:
    #  generic-kernel tolerance applies)
    assert rel_err(run_das(case, kernel=2, t0=t0)[0], run_oracle(case, t0=t0)) <= TOL32
    case = make_case(seq="FSA", interp="linear", seed=35, N=20, I1=70, I2=16)      # 20 % 16 != 0
    assert rel_err(run_das(case, kernel=2)[0], run_oracle(case)) <= TOL32


@pytest.mark.parametrize("interp,prec,fm", [("lanczos3", "single", 0.0), ("cubic", "single", 2.5e6), ("linear", "halfT", 0.0), ("nearest", "single", 0.0), ("cubic", "halfT", 2.5e6)])
def test_reciprocal_mode_with_aperture_weights(interp, prec, fm, monkeypatch):
    """full synthetic aperture with pixel-independent apodization (receive window x transmit window x an N x M table, complex): the two
    traces of an unordered pair share index and interpolation weights but carry DIFFERENT table entries -- w[n, m] and w[m, n]"""
    rng = np.random.default_rng(17)
    case = make_case(seq="FSA", interp=interp, seed=19, N=32, I1=150, I2=19)
    x = case["x"]
    q = (lambda a: a.astype(np.float16).astype(np.float64)) if prec == "halfT" else f32r
    if prec == "halfT":
        x = x.real.astype(np.float16).astype(np.float64) + 1j * x.imag.astype(np.float16).astype(np.float64)
    wn = q(np.hanning(34)[1:-1]).reshape(1, 1, 1, 32, 1)
    wm = q(rng.uniform(0, 1, (1, 1, 1, 1, 32)))
    wm[..., 5] = 0.0
    tab = q(rng.uniform(0.5, 1, (1, 1, 1, 32, 32)))
    tab[..., 7, :] = 0.0
    if prec == "single":
        tab = tab * np.exp(1j * f32r(rng.uniform(0, 1, (1, 1, 1, 32, 32))))
        tab = tab.real.astype(np.float32) + 1j * tab.imag.astype(np.float32)
    fmod = float(np.float32(fm))
    ref = run_oracle(case, apod=(wn, wm, tab), x=x, fmod=fmod)
    out, plan = run_das(case, kernel=2, prec=prec, apod=(wn, wm, tab), fmod=fmod)
    assert plan.kernel == "tiled" and plan.reciprocal and ",wtab" in plan.kernel_name(), plan.kernel_name()
    tol = (2e-3 if interp == "nearest" else 3e-5) if prec == "single" else 3e-3
    assert rel_err(out, ref) <= tol
    monkeypatch.setenv("QDAS_NO_SYM", "1")
    out2, plan2 = run_das(case, kernel=2, prec=prec, apod=(wn, wm, tab), fmod=fmod)
    assert not plan2.reciprocal and rel_err(out2, out) <= tol


@pytest.mark.parametrize("interp", ["nearest", "linear", "cubic", "lanczos3"])
def test_reciprocal_mode_with_half_precision_data(interp, monkeypatch):
    """fp16 channel data in reciprocal mode (launch configuration 8): equals the general fp16 kernel and the oracle on the
    fp16-rounded data; transposed data, the record edge (checked loop) and fmod included"""
    for tpose, fmod in ((False, 0.0), (True, 0.0), (False, 2.5e6)):
        case = make_case(seq="FSA", interp=interp, seed=41, N=32, I1=140, I2=20, data="noise", T=420 if tpose else None)
        xh = (case["x"].real.astype(np.float16).astype(np.float32) + 1j * case["x"].imag.astype(np.float16).astype(np.float32)).astype(np.complex64)
        ref = run_oracle(case, x=xh, fmod=fmod)
        monkeypatch.delenv("QDAS_NO_SYM", raising=False)
        a, pa = run_das(case, kernel=2, prec="halfT", tpose=tpose, fmod=fmod, x=xh)
        assert pa.reciprocal and pa.kernel == "tiled"
        monkeypatch.setenv("QDAS_NO_SYM", "1")
        b, pb = run_das(case, kernel=2, prec="halfT", tpose=tpose, fmod=fmod, x=xh)
        monkeypatch.delenv("QDAS_NO_SYM", raising=False)
        assert not pb.reciprocal
        if interp == "nearest":
            bad = np.abs(a - ref) / np.abs(ref).max() > 3e-3
            assert bad.mean() <= 0.05
        else:
            assert rel_err(a, ref) <= 3e-3 and rel_err(a, b) <= 3e-3      # fp16 output rounding
        assert np.all(a[np.abs(ref) == 0] == 0)


@pytest.mark.parametrize("prec", ["single", "halfT"])
def test_tiled_pixel_receiver_apodization(prec):
    """one I1 x I2 x I3 x N array (acceptance-angle style mask with whole waves of zeros) + pixel-independent arrays"""
    case = make_case(seq="PW", interp="cubic", seed=41, N=12, M=7, I1=150, I2=20)
    rng = np.random.default_rng(4)
    tol = 2e-5 if prec == "single" else 2e-3
    x = case["x"]
    if prec == "halfT":
        x = x.real.astype(np.float16).astype(np.float64) + 1j * x.imag.astype(np.float16).astype(np.float64)
    zz = np.linspace(0, 1, 150)[:, None, None, None, None]
    nn = np.linspace(0, 1, 12)[None, None, None, :, None]
    mask = (np.abs(zz - nn) < 0.35).astype(np.float64) * f32r(rng.uniform(0.5, 1.0, (150, 20, 1, 12, 1)))      # zero for whole depth ranges
    mask = mask.astype(np.float16).astype(np.float64) if prec == "halfT" else mask
    am = f32r(np.hanning(9)[1:-1]).reshape(1, 1, 1, 1, 7)
    am = am.astype(np.float16).astype(np.float64) if prec == "halfT" else am
    ref = run_oracle(case, apod=(mask, am), x=x)
    out, plan = run_das(case, apod=(mask, am), kernel=2, prec=prec)
    assert plan.kernel == "tiled" and plan.fallback_tiles() == 0
    assert rel_err(out, ref) <= tol
    if prec == "single":     # complex pixel weights
        cm = mask * np.exp(1j * f32r(rng.uniform(0, 1, (150, 20, 1, 12, 1))))
        cm = cm.real.astype(np.float32) + 1j * cm.imag.astype(np.float32)
        ref = run_oracle(case, apod=(cm,))
        out, plan = run_das(case, apod=(cm,), kernel=2)
        assert plan.kernel == "tiled" and rel_err(out, ref) <= tol
    # two receive-side arrays are folded into one (test_several_pixel_dependent_arrays_of_one_side_are_folded); an array that depends on the
    # pixel and on both apertures goes to the generic kernel
    from qups_amd import _lib
    out, plan = run_das(case, apod=(mask, mask), kernel=2, prec=prec)
    assert plan.kernel == "tiled" and rel_err(out, run_oracle(case, apod=(mask, mask), x=x)) <= tol
    full = np.broadcast_to(mask, (150, 20, 1, 12, 7)) * f32r(rng.uniform(0.5, 1.0, (1, 1, 1, 12, 7)))      # (not a product of a receive-side and a transmit-side factor)
    full = full.astype(np.float16).astype(np.float64) if prec == "halfT" else f32r(full)
    with pytest.raises(_lib.QdasError, match="pixels x receivers x transmits"):
        run_das(case, apod=(full,), kernel=2, prec=prec)
    out, plan = run_das(case, apod=(full,), kernel=0, prec=prec)
    assert plan.kernel == "generic" and rel_err(out, run_oracle(case, apod=(full,), x=x)) <= max(tol, TOL32)


@pytest.mark.parametrize("seq,prec", [("PW", "single"), ("DV", "halfT"), ("FC", "single")])
def test_full_sum_with_few_transmits_swaps_the_roles_of_the_apertures(seq, prec, monkeypatch):
    """plane-wave compounding with a handful of angles: the 'DAS' sum runs with stage = transmit, block = 32 receivers (fewer, fuller
    stages) -- same image as with the usual roles, weights and per-transmit t0 included"""
    rng = np.random.default_rng(3)
    case = make_case(seq=seq, interp="cubic", seed=23, N=48, M=5, I1=130, I2=21)
    x = case["x"]
    q = (lambda a: a.astype(np.float16).astype(np.float64)) if prec == "halfT" else f32r
    if prec == "halfT":
        x = x.real.astype(np.float16).astype(np.float64) + 1j * x.imag.astype(np.float16).astype(np.float64)
    wn = q(rng.uniform(0.3, 1, (1, 1, 1, 48, 1)))
    wm = q(rng.uniform(0.3, 1, (1, 1, 1, 1, 5)))
    t0 = (case["t0"] + np.float32(1.0 / case["fs"]) * rng.integers(-2, 3, (1, 1, 5))).astype(np.float32).astype(np.float64)
    ref = run_oracle(case, apod=(wn, wm), x=x, t0=t0)
    out, plan = run_das(case, kernel=2, prec=prec, apod=(wn, wm), t0=t0)
    assert plan.kernel == "tiled" and "roles swapped" in plan.kernel_name(), plan.kernel_name()
    tol = (2e-5 if prec == "single" else 2e-3) if seq != "FC" else 1e-4
    assert rel_err(out, ref) <= tol
    monkeypatch.setenv("QDAS_NO_ROLE_SWAP", "1")
    out2, plan2 = run_das(case, kernel=2, prec=prec, apod=(wn, wm), t0=t0)
    assert "roles swapped" not in plan2.kernel_name() and rel_err(out2, out) <= tol


@pytest.mark.parametrize("fun,prec", [("DAS", "single"), ("DAS", "halfT"), ("SYN", "single")])
def test_pixel_only_weights_run_fused(fun, prec):
    """an I1 x I2 weight (a region-of-interest mask, a spatial gain): the same entry for every stage element; tiles outside the region of
    interest have an empty stage list and cost a prologue"""
    rng = np.random.default_rng(14)
    case = make_case(seq="PW", interp="linear", seed=33, N=24, M=7, I1=150, I2=40, xspan=6e-3)
    x = case["x"]
    q = (lambda a: a.astype(np.float16).astype(np.float64)) if prec == "halfT" else f32r
    if prec == "halfT":
        x = x.real.astype(np.float16).astype(np.float64) + 1j * x.imag.astype(np.float16).astype(np.float64)
    roi = q(rng.uniform(0.5, 2.0, (150, 40, 1, 1, 1)))
    roi[:70, :] = 0.0                                                   # (whole tiles of the shallow half)
    roi[:, 30:] = 0.0
    ref = run_oracle(case, fun=fun, apod=(roi,), x=x)
    out, plan = run_das(case, fun=fun, kernel=2, prec=prec, apod=(roi,))
    assert plan.kernel == "tiled", plan.kernel_name()
    assert rel_err(out, ref) <= (2e-5 if prec == "single" else 2e-3)
    assert np.all(out[:70] == 0) and np.all(out[:, 30:] == 0)


@pytest.mark.parametrize("prec", ["single", "halfT"])
@pytest.mark.parametrize("kind", ["two-rx", "depth-by-rx", "roi-by-tx", "two-tx-complex"])
def test_several_pixel_dependent_arrays_of_one_side_are_folded(kind, prec):
    """several pixel-dependent apodization arrays on the receive side (or the transmit side), and arrays that broadcast over a pixel dimension
    (a weight per depth and receiver, I1 x 1 x 1 x N), are multiplied into one plan-owned I x [N | M] array: fused kernel (round 1: generic)"""
    rng = np.random.default_rng(15)
    seq = "PW" if kind in ("two-rx", "depth-by-rx") else "FC"
    case = make_case(seq=seq, interp="cubic", seed=51, N=24, M=10, I1=150, I2=36, xspan=5e-3)
    N, M = case["N"], case["M"]
    x = case["x"]
    q = (lambda a: a.astype(np.float16).astype(np.float64)) if prec == "halfT" else f32r
    if prec == "halfT":
        x = x.real.astype(np.float16).astype(np.float64) + 1j * x.imag.astype(np.float16).astype(np.float64)
    if kind == "two-rx":            # an acceptance mask x a smooth gain, both I1 x I2 x 1 x N
        apod = (q(rng.uniform(0, 1, (150, 36, 1, N, 1)) > 0.35), q(rng.uniform(0.5, 1.5, (150, 36, 1, N, 1))), q(rng.uniform(0.5, 1, (1, 1, 1, N, 1))))
    elif kind == "depth-by-rx":     # aperture growth per depth (no lateral dependence) x a lateral region of interest (pixel-only)
        grow = q(np.abs(np.arange(N) - (N - 1) / 2)[None, None, None, :, None] <= (2 + np.arange(150) * 0.1)[:, None, None, None, None])
        roi = q(np.ones((1, 36, 1, 1, 1))); roi[:, 30:] = 0.0
        apod = (grow, roi)
    elif kind == "roi-by-tx":       # scanline-style transmit weights x a spatial gain
        col = np.arange(36)[None, :, None, None, None]
        apod = (q((np.abs(col - 3 * np.arange(M)[None, None, None, None, :] - 1.5) <= 3.0).astype(np.float64) * np.ones((150, 1, 1, 1, 1))),
                q(rng.uniform(0.5, 2.0, (150, 36, 1, 1, 1))))
    else:                           # two transmit-side arrays, one complex (fp32 only: the half path takes real weights in this harness)
        a2 = rng.uniform(0.5, 1.0, (150, 1, 1, 1, M))
        apod = (q(rng.uniform(0, 1, (150, 36, 1, 1, M)) > 0.4), q(a2) * (1 + 0.5j) if prec == "single" else q(a2))
    ref = run_oracle(case, apod=apod, x=x)
    out, plan = run_das(case, kernel=2, prec=prec, apod=apod)
    assert plan.kernel == "tiled", plan.kernel_name()
    assert rel_err(out, ref) <= (2e-5 if prec == "single" else 3e-3)
    gen, _ = run_das(case, kernel=1, prec=prec, apod=apod)
    assert rel_err(gen, ref) <= (1e-4 if prec == "single" else 3e-3)


@pytest.mark.parametrize("seq,interp", [("FC", "cubic"), ("FC", "nearest"), ("PW", "lanczos3"), ("DV", "linear")])
def test_receive_and_transmit_pixel_arrays_together_run_fused(seq, interp):
    """a transmit-side rule (multiline weights in the reference's shape) AND a receive-side mask (acceptance angle) -- what a focused sequence
    is beamformed with: the transmit is the stage element with its weight, the receive-side product is a second weight per (pixel, block
    element) applied per pair (launch configuration 14, das_tile_impl.h BPIX); pixel-independent arrays join their aperture's product"""
    from qups_amd import apodization as A
    rng = np.random.default_rng(16)
    case = make_case(seq=seq, interp=interp, seed=52, N=24, M=10, I1=150, I2=36, xspan=6e-3)
    N, M = case["N"], case["M"]
    Pv = np.asarray(case["Pv"])
    xv = Pv[0] if seq != "PW" else np.linspace(-3e-3, 3e-3, M)
    tx = f32r(A.ap_multiline(np.linspace(-3e-3, 3e-3, 36), xv))                       # 1 x 36 x 1 x 1 x M
    nrm = np.asarray(G.linear_array(N, 0.3e-3)[1], np.float32).astype(np.float64)
    rx = f32r(A.ap_acceptance_angle(case["Pi"], case["Pr"], nrm, 30.0))                # 150 x 36 x 1 x N x 1
    gain = f32r(rng.uniform(0.5, 1.5, (150, 1, 1, N, 1)))                              # a second receive-side array (per depth)
    wn = f32r(rng.uniform(0.3, 1, (1, 1, 1, N, 1)))
    wm = f32r(rng.uniform(0.3, 1, (1, 1, 1, 1, M))); wm[..., 3] = 0.0
    apod = (tx, rx, gain, wn, wm)
    ref = run_oracle(case, apod=apod)
    out, plan = run_das(case, kernel=2, apod=apod)
    assert plan.kernel == "tiled" and "W=384" in plan.kernel_name() and "roles swapped" in plan.kernel_name(), plan.kernel_name()
    if interp == "nearest":
        bad = np.abs(out - ref) / np.abs(ref).max() > 1e-4
        assert bad.mean() <= 0.02
    else:
        assert rel_err(out, ref) <= 2e-5
    assert np.all(out[np.abs(ref) == 0] == 0)
    gen, _ = run_das(case, kernel=1, apod=apod)
    assert rel_err(gen, ref) <= 1e-4 or interp == "nearest"
    # a dead channel under a zero receive-side weight stays out (src/bf.cu:122,126): non-finite samples of receiver 5 wherever rx == 0
    if interp == "cubic":
        x2 = np.array(case["x"], dtype=np.complex128)
        rx2 = rx.copy(); rx2[:, :, :, 5, :] = 0.0
        x2[:, 5, :] = np.nan
        out2, _ = run_das(case, kernel=2, apod=(tx, rx2, gain, wn, wm), x=x2)
        x3 = np.array(case["x"], dtype=np.complex128); x3[:, 5, :] = 0.0
        ref3 = run_oracle(case, apod=(tx, rx2, gain, wn, wm), x=x3)
        assert np.isfinite(out2).all() and rel_err(out2, ref3) <= 2e-5


def test_two_sided_pixel_weights_outside_the_fused_scope_use_the_generic_kernel():
    """fp16 data, complex weights, remodulation or an N x M table next to two-sided pixel weights: generic kernel (same results)"""
    rng = np.random.default_rng(17)
    case = make_case(seq="FC", interp="linear", seed=53, N=16, M=6, I1=80, I2=20)
    rx = f32r(rng.uniform(0, 1, (80, 20, 1, 16, 1)) > 0.3)
    tx = f32r(rng.uniform(0, 1, (80, 20, 1, 1, 6)))
    nm = f32r(rng.uniform(0.5, 1, (1, 1, 1, 16, 6)))
    from qups_amd import _lib
    with pytest.raises(_lib.QdasError, match="together run fused for"):
        run_das(case, kernel=2, apod=(rx, tx, nm))
    ref = run_oracle(case, apod=(rx, tx, nm))
    out, plan = run_das(case, kernel=0, apod=(rx, tx, nm))
    assert plan.kernel == "generic" and rel_err(out, ref) <= 1e-4
    out, plan = run_das(case, kernel=0, apod=(rx, tx * (1 + 0.5j)))
    assert plan.kernel == "generic" and rel_err(out, run_oracle(case, apod=(rx, tx * (1 + 0.5j)))) <= 1e-4


def test_mul_mode_with_pixel_by_transmit_weights():
    """'MUL' (keep the transmit dimension) with a weight per (pixel, transmit): the transmit is the stage element of that mode, the weight
    scales its plane"""
    rng = np.random.default_rng(13)
    case = make_case(seq="DV", interp="linear", seed=43, N=20, M=6, I1=120, I2=18)
    wt = f32r(rng.uniform(0, 1, (120, 18, 1, 1, 6)) * (rng.uniform(0, 1, (120, 18, 1, 1, 6)) > 0.3))
    ref = run_oracle(case, fun="MUL", apod=(wt,))
    out, plan = run_das(case, fun="MUL", kernel=2, apod=(wt,))
    assert plan.kernel == "tiled"
    assert out.shape == ref.shape and rel_err(out, ref) <= 2e-5


@pytest.mark.parametrize("prec", ["single", "halfT"])
def test_pixel_by_transmit_apodization_runs_fused_with_swapped_roles(prec):
    """a weight per (pixel, transmit) -- the scanline / multiline / parallelogram transmit apodization of focused sequences, I1 x I2 x 1 x 1 x M
    (reference src/UltrasoundSystem.m:4892-5074) -- is a pixel x stage-element weight once the transmit is the stage element: fused kernel,
    stage list of the transmits that matter to the tile; plus a receive window (pixel-independent)"""
    rng = np.random.default_rng(12)
    case = make_case(seq="FC", interp="cubic", seed=29, N=32, M=12, I1=140, I2=36, xspan=5e-3)
    x = case["x"]
    q = (lambda a: a.astype(np.float16).astype(np.float64)) if prec == "halfT" else f32r
    if prec == "halfT":
        x = x.real.astype(np.float16).astype(np.float64) + 1j * x.imag.astype(np.float16).astype(np.float64)
    col = np.arange(36)[None, :, None, None, None]
    mm = np.arange(12)[None, None, None, None, :]
    scan = (np.abs(col / 3.0 - mm) <= 1.0) * q(rng.uniform(0.5, 1.0, (140, 36, 1, 1, 12)))      # each image column listens to its 2-3 nearest transmits
    wn = q(np.hanning(34)[1:-1]).reshape(1, 1, 1, 32, 1)
    ref = run_oracle(case, apod=(scan, wn), x=x)
    out, plan = run_das(case, kernel=2, prec=prec, apod=(scan, wn))
    assert plan.kernel == "tiled" and "roles swapped" in plan.kernel_name(), plan.kernel_name()
    assert rel_err(out, ref) <= (1e-4 if prec == "single" else 2e-3)
    gen, gplan = run_das(case, kernel=1, prec=prec, apod=(scan, wn))
    assert rel_err(out, gen) <= (2e-4 if prec == "single" else 3e-3)


@pytest.mark.parametrize("seq", ["PW", "FSA"])
def test_thousand_element_apertures_stay_on_the_fused_kernel(seq):
    """a 32 x 32 matrix array's worth of elements on both sides (N = M = 1024, not reciprocal): the LDS header of the tile (window
    bases, receiver records, transmit tables) plus two window buffers still fit one CU; checked against the generic kernel"""
    torch = _torch()
    from qups_amd import DasPlan, build_problem, parse_options
    from qups_amd.das_spec import _colmajor
    case = make_case(seq=seq, interp="linear", seed=41, N=1024, M=1024, I1=64, I2=16, pitch=0.05e-3, T=8, data="noise", zlim=(3e-3, 5e-3))
    g0 = torch.Generator(device="cuda").manual_seed(41)                    # (160 x 1024 x 1024 samples: drawn on the device)
    xt = torch.view_as_complex(torch.randn((160, 1024, 1024, 2), generator=g0, device="cuda", dtype=torch.float32))
    po = parse_options(xt, list(case["opt"]) + ["interp", "linear"])
    prob = build_problem("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], tuple(xt.shape), case["t0"], case["fs"], case["c"], po)
    xc = _colmajor(xt)
    tiled = DasPlan(prob, kernel=2)
    assert tiled.kernel == "tiled" and tiled.reciprocal == (seq == "FSA")
    y = tiled.execute_colmajor(xc).cpu().numpy()
    g = DasPlan(prob, kernel=1).execute_colmajor(xc).cpu().numpy()
    # white noise and 10^6 pairs per pixel, a record that ends inside the image: the edge rule is a step in tau, so a pair within
    # rounding of the record's end is counted by one kernel and not by the other -- ONE sample of a million, in a few pixels
    e = np.abs(y - g).reshape(-1) / np.abs(g).max()
    assert np.abs(g).max() > 0 and np.median(e) <= 2e-5 and (e > 3e-4).mean() <= 0.02 and e.max() <= 5e-3, (np.median(e), (e > 3e-4).mean(), e.max())


@pytest.mark.parametrize("fun", ["DAS", "SYN"])
@pytest.mark.parametrize("ks", [1, 3])
def test_receivers_without_weight_in_a_tile_are_dropped_from_its_stage_list(fun, ks, monkeypatch):
    """a pixel x receiver mask: receivers 0-4 weigh nothing anywhere, the shallow third of the image weighs nothing at all (tiles with an
    EMPTY stage list), the rest follows a depth-dependent aperture -- the tile's stage list holds the active receivers only
    (das_tile_impl.h plan_stages); 'SYN' planes of dropped receivers stay zero; with and without an aperture split"""
    monkeypatch.setenv("QDAS_KSPLIT", str(ks))
    case = make_case(seq="PW", interp="cubic", seed=31, N=24, M=9, I1=200, I2=21, zlim=(4e-3, 24e-3))
    rng = np.random.default_rng(8)
    zz = np.linspace(0, 1, 200)[:, None, None, None, None]
    nn = np.arange(24)[None, None, None, :, None]
    mask = ((np.abs(nn - 12) <= 2 + 14 * zz) & (nn >= 5) & (zz > 0.34)) * f32r(rng.uniform(0.5, 1.0, (200, 21, 1, 24, 1)))
    assert (mask[:66] == 0).all() and (mask[..., :5, :] == 0).all() and (mask != 0).any()
    ref = run_oracle(case, fun=fun, apod=(mask,))
    out, plan = run_das(case, fun=fun, apod=(mask,), kernel=2)
    assert plan.kernel == "tiled" and plan.fallback_tiles() == 0
    assert rel_err(out, ref) <= 2e-5
    assert np.all(out[:64] == 0)                                    # (whole tiles of the shallow third)
    if fun == "SYN":
        assert np.all(out[..., :5, :] == 0)


@pytest.mark.parametrize("tz,wz", [(64, 64), (64, 8), (64, 4), (32, 32), (32, 4), (16, 16), (16, 8), (8, 8), (8, 4)])
@pytest.mark.parametrize("seq,prec", [("FSA", "single"), ("PW", "single"), ("DV", "halfT")])
def test_tile_shapes_forced(tz, wz, seq, prec, monkeypatch):
    """every tile footprint (64x16 ... 8x128 pixels) and wave footprint (64x1 ... 4x16) gives the same image, ragged edges included"""
    monkeypatch.setenv("QDAS_TILE_Z", str(tz))
    monkeypatch.setenv("QDAS_WAVE_Z", str(wz))
    case = make_case(seq=seq, interp="lanczos3", seed=21, N=16, I1=150, I2=37, zlim=(4e-3, 14e-3), xspan=3e-3)
    xq = case["x"]
    if prec == "halfT":
        xq = xq.real.astype(np.float16).astype(np.float64) + 1j * xq.imag.astype(np.float16).astype(np.float64)
    ref = run_oracle(case, x=xq)
    out, plan = run_das(case, kernel=2, prec=prec)
    assert plan.kernel == "tiled" and plan.tile_shape() == (tz, 16 * 64 // tz) and plan.wave_shape() == (wz, 64 // wz)
    assert plan.fallback_tiles() == 0
    assert rel_err(out, ref) <= (2e-3 if prec == "halfT" else TOL32)


def test_tile_shape_follows_the_axial_pitch():
    """coarse axial sampling (5.5 samples of delay per pixel): the plan picks a shorter, wider tile instead of falling back"""
    fine = make_case(seq="FSA", interp="cubic", seed=22, N=16, I1=200, I2=40, zlim=(4e-3, 14e-3), xspan=2e-3)
    coarse = make_case(seq="FSA", interp="cubic", seed=22, N=16, I1=48, I2=40, zlim=(4e-3, 14e-3), xspan=2e-3)
    out_f, plan_f = run_das(fine, kernel=2)
    out_c, plan_c = run_das(coarse, kernel=2)
    assert plan_f.tile_shape() == (64, 16) and plan_f.wave_shape()[0] <= 16     # 2.6 samples/pixel: shallow waves (LDS banks)
    assert plan_c.tile_shape()[0] < 64
    assert plan_f.fallback_tiles() == 0 and plan_c.fallback_tiles() == 0
    assert rel_err(out_f, run_oracle(fine)) <= TOL32
    assert rel_err(out_c, run_oracle(coarse)) <= TOL32


@pytest.mark.parametrize("ks", [2, 3, 4, 8])
@pytest.mark.parametrize("seq,N,prec,mask", [("FSA", 128, "single", False), ("FSA", 72, "single", False), ("FSA", 64, "halfT", False), ("PW", 16, "single", True),
                                             ("DV", 16, "halfT", True)])
def test_aperture_split(ks, seq, N, prec, mask, monkeypatch):
    """ksplit workgroups per tile (reciprocal mode: interleaved transmit blocks -- of 32 transmits on the reciprocity-folded frame, the last one
    partial at N = 72; fp16 data: folded into complex64 --; otherwise receiver ranges) + fixed-order reduce"""
    monkeypatch.setenv("QDAS_KSPLIT", str(ks))
    geo = dict(pitch=0.1e-3, zlim=(10e-3, 16e-3)) if N > 32 else dict(zlim=(4e-3, 14e-3))
    case = make_case(seq=seq, interp="lanczos3", seed=31, N=N, I1=100, I2=21, xspan=3e-3, **geo)
    xq = case["x"]
    apod = ()
    if mask:
        rng = np.random.default_rng(5)
        a = (rng.uniform(0, 1, (100, 21, 1, N, 1)) > 0.5).astype(np.float64)
        a[:, 3:9] = 0.0                                        # whole waves without weight
        apod = (a,)
    if prec == "halfT":
        xq = xq.real.astype(np.float16).astype(np.float64) + 1j * xq.imag.astype(np.float16).astype(np.float64)
    ref = run_oracle(case, x=xq, apod=apod)
    out, plan = run_das(case, kernel=2, prec=prec, apod=apod)
    cap = min(8, (N + 31) // 32 if seq == "FSA" else N)              # (fp16 reciprocal data run the folded fp32 kernels too)
    auto = 1 << (cap.bit_length() - 1)                         # tiny image: the plan itself splits as far as it may
    assert plan.aperture_split() == (ks if ks <= cap else auto)
    assert plan.fallback_tiles() == 0
    assert rel_err(out, ref) <= (2e-3 if prec == "halfT" else TOL32)
    # deterministic: a second run is bit-identical
    out2, _ = run_das(case, kernel=2, prec=prec, apod=apod)
    assert np.array_equal(out, out2)


@pytest.mark.parametrize("seq", ["FSA", "PW", "FC"])
def test_large_time_offset(seq):
    """t0 = -100 us (2000 samples at 20 MHz): the fp32 window-base estimates of the tiled kernel's prologue cancel two large
    numbers; their error margin must still cover the fp64 delays (no fallback, same image)"""
    case = make_case(seq=seq, interp="lanczos3", seed=61, N=16, I1=150, I2=20, zlim=(4e-3, 16e-3), xspan=3e-3, t0=float(np.float32(-1.0e-4)))
    assert case["T"] > 2000
    ref = run_oracle(case)
    out, plan = run_das(case, kernel=2)
    assert plan.kernel == "tiled"
    if seq == "FC":
        assert plan.fallback_tiles() <= 4 and rel_err(out, ref) <= TOL32
    else:
        assert plan.fallback_tiles() == 0 and rel_err(out, ref) <= 3e-5


@pytest.mark.parametrize("seq,interp,prec,extra", [("PW", "cubic", "single", {}), ("FSA", "lanczos3", "single", {"N": 24}), ("DV", "linear", "halfT", {}),
                                                   ("PW", "nearest", "single", {}), ("FC", "cubic", "single", {"fmod": 2.0e6}),
                                                   ("PW", "cubic", "single", {"wtab": True}), ("DV", "cubic", "halfT", {"wpix": True}),
                                                   ("PW", "lanczos3", "single", {"ks": 4})])
@pytest.mark.parametrize("F", [2, 3, 4, 7])
def test_frame_pairs_share_one_launch(seq, interp, prec, extra, F, monkeypatch):
    """F frames through one plan: the tiled kernel beamforms them two at a time with shared tap indices / weights; every frame must
    equal what a frame-by-frame run (QDAS_NO_FB2=1) and the oracle give"""
    import torch
    from qups_amd import DasPlan, build_problem, parse_options
    from qups_amd.das_spec import _cast_data, _colmajor
    from oracle import das_oracle as O
    N = extra.get("N", 16)
    case = make_case(seq=seq, interp=interp, seed=71, N=N, I1=150, I2=21, zlim=(4e-3, 15e-3), xspan=3e-3, data="noise")
    M = case["M"]
    rng = np.random.default_rng(72)
    xs = np.stack([case["x"]] + [(rng.standard_normal(case["x"].shape) + 1j * rng.standard_normal(case["x"].shape)).astype(np.complex64)
                                 for _ in range(F - 1)], axis=3)                     # T x N x M x F
    if prec == "halfT":
        xs = (xs.real.astype(np.float16).astype(np.float32) + 1j * xs.imag.astype(np.float16).astype(np.float32)).astype(np.complex64)
    apod = []
    q = (lambda a: a.astype(np.float16).astype(np.float64)) if prec == "halfT" else (lambda a: a.astype(np.float32).astype(np.float64))
    if extra.get("wtab"):
        a = q(rng.uniform(0, 1, (1, 1, 1, 1, M)))
        a[..., 1] = 0.0
        apod.append(a * (1 + 0.5j))
    if extra.get("wpix"):
        a = q(rng.uniform(0, 1, (150, 21, 1, N, 1)) > 0.4)
        a[:40] = 0.0
        apod.append(a)
    fmod = float(np.float32(extra.get("fmod", 0.0)))
    if "ks" in extra:
        monkeypatch.setenv("QDAS_KSPLIT", str(extra["ks"]))
    opts = list(case["opt"]) + ["interp", interp, "input-precision", prec, "modulation", fmod]
    for a in apod:
        opts += ["apod", a]
    xt = torch.from_numpy(np.ascontiguousarray(xs))
    prob = build_problem("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], tuple(xt.shape), case["t0"], case["fs"], case["c"],
                         parse_options(xt, opts))
    xc = _colmajor(_cast_data(xt, prob.prec, torch.device("cuda:0")))

    def run():
        plan = DasPlan(prob, kernel=2)
        y = plan.execute_colmajor(xc, F)
        torch.cuda.synchronize()
        return y.to(torch.complex64).cpu().numpy().reshape(F, -1), plan

    ya, plan = run()
    monkeypatch.setenv("QDAS_NO_FB2", "1")
    yb, _ = run()
    monkeypatch.delenv("QDAS_NO_FB2")
    assert plan.kernel == "tiled"
    tol = 3e-3 if prec == "halfT" else 3e-5
    for f in range(F):
        ref = O.das_spec("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], xs[..., f], case["t0"], case["fs"], cinv_f32(case["c"]),
                         VS=case["VS"], DV=case["DV"], interp=interp, apod=tuple(apod), fmod=fmod).reshape(-1, order="F")
        den = np.abs(ref).max()
        if interp == "nearest":
            assert np.mean(np.abs(ya[f] - ref) / den > 1e-4) < 0.05
            assert np.mean(np.abs(ya[f] - yb[f]) / den > 1e-4) < 0.05
        elif seq == "FC":
            assert np.abs(ya[f] - ref).max() / den <= 1e-4 and np.abs(ya[f] - yb[f]).max() / den <= 1e-4      # focal-depth tiles: generic kernel
        else:
            assert np.abs(ya[f] - ref).max() / den <= tol, (f, np.abs(ya[f] - ref).max() / den)
            assert np.abs(ya[f] - yb[f]).max() / den <= tol


@pytest.mark.parametrize("seq,interp,extra", [("PW", "cubic", {}), ("FSA", "lanczos3", {}), ("DV", "linear", {"wtab": True}),
                                              ("PW", "cubic", {"wpix": True}), ("PW", "lanczos3", {"ks": 4}), ("FC", "cubic", {"fmod": 2.0e6}),
                                              ("PW", "cubic", {"F": 3}), ("PW", "linear", {"shard": True})])
@pytest.mark.parametrize("fun", ["SYN", "MUL"])
def test_syn_mode_in_the_tiled_kernel(fun, seq, interp, extra, monkeypatch):
    """'SYN' (keep the receive dimension, kern/das_spec.m:263-269): I x N planes accumulated by the tiled kernel, with weight tables,
    pixel x receiver masks, a split aperture, frames, shards writing into a full-size buffer"""
    import torch
    from qups_amd import DasPlan, build_problem, parse_options
    from qups_amd.das_spec import _cast_data, _colmajor
    from oracle import das_oracle as O
    if fun == "MUL" and extra.get("wpix"):
        pytest.skip("'MUL' with a pixel x receiver array runs the generic kernel")
    N = 16
    F = extra.get("F", 1)
    case = make_case(seq=seq, interp=interp, seed=81, N=N, I1=150, I2=21, zlim=(4e-3, 15e-3), xspan=3e-3, data="noise")
    M = case["M"]
    rng = np.random.default_rng(82)
    xs = np.stack([case["x"]] + [(rng.standard_normal(case["x"].shape) + 1j * rng.standard_normal(case["x"].shape)).astype(np.complex64)
                                 for _ in range(F - 1)], axis=3)
    apod = []
    if extra.get("wtab"):
        a = f32r(rng.uniform(0, 1, (1, 1, 1, 1, M)))
        a[..., 1] = 0.0
        apod.append(a * (1 + 0.5j))
        apod.append(f32r(rng.uniform(0.5, 1, (1, 1, 1, N, 1))))
    if extra.get("wpix"):
        a = f32r(rng.uniform(0, 1, (150, 21, 1, N, 1)) > 0.4)
        a[:40] = 0.0
        apod.append(a)
    fmod = float(np.float32(extra.get("fmod", 0.0)))
    if "ks" in extra:
        monkeypatch.setenv("QDAS_KSPLIT", str(extra["ks"]))
    opts = list(case["opt"]) + ["interp", interp, "modulation", fmod]
    for a in apod:
        opts += ["apod", a]
    xt = torch.from_numpy(np.ascontiguousarray(xs))
    prob = build_problem(fun, case["Pi"], case["Pr"], case["Pv"], case["Nv"], tuple(xt.shape), case["t0"], case["fs"], case["c"],
                         parse_options(xt, opts))
    xc = _colmajor(_cast_data(xt, prob.prec, torch.device("cuda:0")))
    I = 150 * 21
    kw = dict(i_begin=I // 3, i_count=I // 2) if extra.get("shard") else {}
    plan = DasPlan(prob, kernel=2, **kw)
    y = plan.execute_colmajor(xc, F)                               # (F, [M], [N], count)
    torch.cuda.synchronize()
    P = N if fun == "SYN" else M                                   # planes: one per receiver / per transmit
    assert plan.kernel == "tiled" and tuple(y.shape)[:3] == ((F, 1, N) if fun == "SYN" else (F, M, 1))
    out = y.cpu().numpy().reshape(F, P, -1)
    for f in range(F):
        ref = O.das_spec(fun, case["Pi"], case["Pr"], case["Pv"], case["Nv"], xs[..., f], case["t0"], case["fs"], cinv_f32(case["c"]),
                         VS=case["VS"], DV=case["DV"], interp=interp, apod=tuple(apod), fmod=fmod)          # I1 x I2 x 1 x N x 1
        ref = ref.reshape(I, P, order="F")
        if kw:
            ref = ref[kw["i_begin"]: kw["i_begin"] + kw["i_count"]]
        got = out[f].T                                             # count x planes
        assert np.abs(got - ref).max() / np.abs(ref).max() <= (1e-4 if seq == "FC" else 3e-5)


@pytest.mark.parametrize("fun,ndev,mem", [("DAS", 2, "device"), ("DAS", 3, "device"), ("SYN", 2, "device"), ("BF", 2, "device"),
                                          ("DAS", 2, "host"), ("MUL", 3, "host")])
def test_sharded_c_abi_entry_on_one_device(fun, ndev, mem):
    """qdas_plan_create_sharded / _execute_sharded (one host thread, N streams): with every shard on device 0 the slabs, the
    replication bookkeeping and the plane-wise gather are exercised and the image must equal the single-plan image bit for bit"""
    import ctypes as C
    import torch
    from qups_amd import DasPlan, MultiDevicePlan, _lib, build_problem, parse_options
    case = make_case(seq="PW", interp="cubic", seed=41, N=10, M=6, I1=203, I2=11)      # I = 2233: ragged slabs
    x = torch.from_numpy(case["x"])
    opts = parse_options(x, list(case["opt"]) + ["interp", "cubic"])
    T, N, M = case["x"].shape
    prob = build_problem(fun, case["Pi"], case["Pr"], case["Pv"], case["Nv"], (T, N, M), case["t0"], case["fs"], case["c"], opts)
    one = DasPlan(prob, mirror=False)       # (slabs run the plain kernel; the lateral-mirror mode of a whole-image plan sums the mirrored half in another order)
    y1 = one.feval(x)
    if mem == "device":
        mp = MultiDevicePlan(prob, devices=[0] * ndev)
        sh = mp.shards()
        assert len(sh) == ndev and sum(c for _, _, c, _ in sh) == prob.I and [b for _, b, _, _ in sh] == [prob.I * g // ndev for g in range(ndev)]
        y2 = mp.feval(x)
        y3 = mp.feval(x)                                  # plan reuse
        torch.cuda.synchronize()
        assert torch.equal(y2, y1) and torch.equal(y3, y1)
        mp.close()
    else:                                                 # host-resident inputs through the same entry (what the MEX gateway passes)
        L = _lib.lib()
        d = _lib.Desc()
        keep = [np.ascontiguousarray(a) for a in (prob.Pi, prob.Pr, prob.Pv, prob.Nv, prob.cinv)]
        acs = (C.c_uint64 * len(prob.acstride))(*[int(v) for v in prob.acstride])
        d.sz = _lib.Sizes(prob.T, prob.N, prob.M, *prob.Isz, prob.S, prob.flag, int(prob.VS), int(prob.DV), 1)
        d.fs, d.fmod = prob.fs, prob.fmod
        d.Pi, d.Pr, d.Pv, d.Nv, d.cinv = (a.ctypes.data for a in keep)
        d.acstride, d.mem, d.kernel, d.device = acs, _lib.MEM_HOST, 0, 0
        h = C.c_void_p()
        devs = (C.c_int * ndev)(*([0] * ndev))
        _lib.check(L.qdas_plan_create_sharded(C.byref(h), C.byref(d), ndev, devs))
        xh = np.ascontiguousarray(case["x"].astype(np.complex64).transpose(2, 1, 0))        # column-major T x N x M
        oN, oM = prob.osize
        yh = np.empty((oM, oN, prob.I), np.complex64)
        _lib.check(L.qdas_plan_execute_sharded(h, xh.ctypes.data, yh.ctypes.data, None))
        L.qdas_plan_destroy_sharded(h)
        assert np.array_equal(yh.transpose(2, 1, 0), y1.cpu().numpy())
    one.close()


@pytest.mark.parametrize("seq,interp,extra", [("PW", "cubic", {}), ("FSA", "lanczos3", {}), ("FC", "linear", {"fmod": 3e6}), ("DV", "nearest", {}),
                                              ("PW", "linear", {"apod": True}), ("FSA", "cubic", {"tpose": True})])
def test_bf_mode_on_the_tiled_kernel(seq, interp, extra):
    """'BF' (keep both aperture dimensions, src/bf.cu:134-135) runs on the fused kernel for fp32 data: every pair's weighted sample
    goes to plane nm of y.  Against the float64 oracle and against the one-pixel-per-lane kernel; ragged tiles, odd M, record edges."""
    import torch
    from oracle import das_oracle as O
    from qups_amd import das_spec
    case = make_case(seq=seq, interp=interp, seed=17, N=9, M=7 if seq != "FSA" else None, I1=70, I2=19)
    x = case["x"]
    N, M = x.shape[1], x.shape[2]
    fmod = float(np.float32(extra.get("fmod", 0.0)))
    opts = list(case["opt"]) + ["interp", interp, "modulation", fmod]
    apod = []
    if extra.get("apod"):
        rng = np.random.default_rng(2)
        a = (rng.random((1, 1, 1, N, M)) + 1j * rng.random((1, 1, 1, N, M))).astype(np.complex64)
        a[0, 0, 0, 2, :] = 0                                     # a dead receiver: its planes must hold exact zeros
        apod = [a]
        opts += ["apod", a]
    xin = x
    if extra.get("tpose"):
        xin = np.ascontiguousarray(np.swapaxes(x, 1, 2))
        opts += ["transpose", True]
    args = (case["Pi"], case["Pr"], case["Pv"], case["Nv"], torch.from_numpy(xin), case["t0"], case["fs"], case["c"])
    y2, p2 = das_spec("BF", *args, *opts, return_plan=True, kernel=2)
    y1, p1 = das_spec("BF", *args, *opts, return_plan=True, kernel=1)
    assert p2.kernel == "tiled" and p1.kernel == "generic"
    ref = O.das_spec("BF", case["Pi"], case["Pr"], case["Pv"], case["Nv"], xin, case["t0"], case["fs"], cinv_f32(case["c"]),
                     VS=case["VS"], DV=case["DV"], interp=interp, apod=apod, fmod=fmod, tpose=bool(extra.get("tpose")))
    y2n, y1n = y2.cpu().numpy().reshape(ref.shape), y1.cpu().numpy().reshape(ref.shape)
    tol = 1e-2 if interp == "nearest" else (2e-4 if fmod else 2e-5)
    assert rel_err(y2n, ref) <= tol
    assert rel_err(y2n, y1n) <= max(tol, 1e-4)
    if extra.get("apod"):
        assert not np.any(y2n[:, :, :, 2, :])
    auto, pa = das_spec("BF", *args, *opts, return_plan=True)
    assert pa.kernel == "tiled"                                   # the fused kernel is the default for fp32 'BF'
