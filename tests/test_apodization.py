"""Receive-apodization generators: host restatement (qups_amd.apodization) vs the oracle's line-by-line restatement of the
reference (its degree / atan2d formulation), and -- on the GPU -- weights generated inside the kernels vs the materialised arrays."""
from __future__ import annotations

import numpy as np
import pytest

from qups_amd import geometry as G
from tests.cases import cinv_f32, make_case, rel_err


def _geom(convex, seed=0, I1=37, I2=23):
    if convex:
        Pr, nrm = G.convex_array(24, 40e-3, 0.6)
        ang = np.rad2deg(np.arctan2(nrm[0], nrm[2]))
        Pi = G.scan_polar(np.linspace(40e-3, 90e-3, I1), np.linspace(-30, 30, I2), origin=np.array([0.0, 0.0, -40e-3]))
    else:
        Pr, nrm = G.linear_array(24, 0.3e-3)
        ang = np.zeros(24)
        Pi = G.scan_cartesian(np.linspace(-6e-3, 6e-3, I2), np.linspace(1e-3, 25e-3, I1))
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    return f32(Pi), f32(Pr), f32(nrm), ang


@pytest.mark.parametrize("convex", [False, True])
def test_host_generators_match_the_reference_formulation(convex):
    from oracle import das_oracle as O
    from qups_amd import apodization as A
    Pi, Pr, nrm, ang = _geom(convex)
    for th in (20.0, 45.0, 60.0):
        a, b = A.ap_acceptance_angle(Pi, Pr, nrm, th), O.ap_acceptance_angle(Pi, Pr, nrm, th)
        assert a.shape == (37, 23, 1, 24, 1) and 0.05 < a.mean() < 0.999
        assert np.mean(a != b) <= 1e-3                     # step function: only points on the step may differ
        c, d = A.ap_cosine_angle(Pi, Pr, nrm, th), O.ap_cosine_angle(Pi, Pr, nrm, th)
        assert np.abs(c - d).max() <= 1e-12
    for f, D in ((1.0, np.inf), (1.5, 6e-3), (2.5, np.inf)):
        a = A.ap_aperture_growth(Pi, Pr, nrm, f, D)
        b = O.ap_aperture_growth(Pi, Pr, ang if convex else None, f, D)
        assert a.shape == b.shape and 0.02 < a.mean() < 0.98
        assert np.mean(a != b) <= 2e-3                     # dot/cross form vs the reference's atan2d/sind/cosd form
    assert A.rx_apod_spec("fnumber", normals=nrm)["kind"] == (4 if convex else 3)


def test_cosd_is_exact_like_matlab():
    from qups_amd.apodization import cosd
    assert cosd(90) == 0.0 and cosd(60) == 0.5 and cosd(0) == 1.0 and cosd(180) == -1.0
    assert abs(cosd(45) - np.sqrt(0.5)) < 1e-16
    # element on a pixel: 0/0 -> NaN -> outside the acceptance cone, cosine weight 1 (MATLAB max/min ignore NaN)
    from qups_amd import apodization as A
    Pi = np.zeros((3, 1, 1, 1)); Pr = np.zeros((3, 1)); nrm = np.array([[0.0], [0.0], [1.0]])
    assert A.ap_acceptance_angle(Pi, Pr, nrm, 45).ravel()[0] == 0.0
    assert A.ap_cosine_angle(Pi, Pr, nrm, 45).ravel()[0] == 1.0


def test_rx_apod_option_validation():
    import torch
    from qups_amd import DasError, build_problem, parse_options
    case = make_case(seq="FSA", N=8, I1=8, I2=4)
    x = torch.from_numpy(case["x"])
    with pytest.raises(DasError):
        build_problem("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], tuple(x.shape), case["t0"], case["fs"], case["c"],
                      parse_options(x, list(case["opt"]) + ["rx-apod", "acceptance"]))
    with pytest.raises(DasError):
        build_problem("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], tuple(x.shape), case["t0"], case["fs"], case["c"],
                      parse_options(x, list(case["opt"]) + ["rx-apod", dict(kind=1, p=(0.5, 0.0), normals=np.zeros((3, 5)))]))


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [1, 2])
@pytest.mark.parametrize("kind,kw", [("acceptance", dict(theta=25.0)), ("cosine", dict(theta=35.0)), ("fnumber", dict(f=1.2, Dmax=4e-3))])
@pytest.mark.parametrize("convex,prec", [(False, "single"), (True, "single"), (True, "halfT")])
def test_generated_weights_equal_the_materialised_array(kernel, kind, kw, convex, prec):
    import torch
    from qups_amd import apodization as A, das_spec
    from oracle import das_oracle as O
    case = make_case(seq="DV" if convex else "FSA", interp="cubic", seed=41, N=16, M=None if not convex else 5, I1=120, I2=21,
                     convex=convex, zlim=(4e-3, 16e-3), xspan=5e-3)
    nrm = G.convex_array(16, 40e-3, 0.6)[1] if convex else G.linear_array(16, 0.3e-3)[1]
    nrm = np.asarray(nrm, np.float32).astype(np.float64)
    x = case["x"]
    if prec == "halfT":
        x = (x.real.astype(np.float16).astype(np.float32) + 1j * x.imag.astype(np.float16).astype(np.float32)).astype(np.complex64)
    spec = A.rx_apod_spec(kind, normals=nrm, **kw)
    arr = {"acceptance": lambda: A.ap_acceptance_angle(case["Pi"], case["Pr"], nrm, kw["theta"]),
           "cosine": lambda: A.ap_cosine_angle(case["Pi"], case["Pr"], nrm, kw["theta"]),
           "fnumber": lambda: A.ap_aperture_growth(case["Pi"], case["Pr"], nrm, kw["f"], kw["Dmax"])}[kind]()
    assert 0.02 < float((arr > 1e-9).mean()) < 0.999           # the rule cuts a real part of the aperture
    ref = O.das_spec("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], x, case["t0"], case["fs"], cinv_f32(case["c"]),
                     VS=case["VS"], DV=case["DV"], interp="cubic", apod=(arr,))
    opts = list(case["opt"]) + ["interp", "cubic", "input-precision", prec]
    xt = torch.from_numpy(x)
    yg, plan = das_spec("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], xt, case["t0"], case["fs"], case["c"], *opts,
                        "rx-apod", spec, return_plan=True, kernel=kernel)
    arr_q = arr.astype(np.float16).astype(np.float64) if prec == "halfT" else arr
    ya = das_spec("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], xt, case["t0"], case["fs"], case["c"], *opts,
                  "apod", arr_q, kernel=kernel)
    torch.cuda.synchronize()
    assert plan.kernel == ("generic" if kernel == 1 else "tiled")
    g, a = yg.to(torch.complex64).cpu().numpy(), ya.to(torch.complex64).cpu().numpy()
    tol = 3e-3 if prec == "halfT" else 1e-4
    assert rel_err(g, ref) <= tol
    if kind == "cosine" and prec == "halfT":
        assert rel_err(g, a) <= 3e-3                       # the materialised fp16 array rounds the smooth weights
    else:
        assert rel_err(g, a) <= (2e-3 if prec == "halfT" else 2e-6)       # same weights, same kernel: rounding-level agreement


def test_transmit_side_rules_shapes_and_values():
    """``ap_scanline`` / ``ap_multiline`` / ``ap_translating_aperture`` (reference src/UltrasoundSystem.m:4892-5163): broadcast shapes
    (singleton over depth), partition of unity between neighbouring transmits, known values on a hand-checked grid"""
    from qups_amd import apodization as A
    xi = np.linspace(-4e-3, 4e-3, 17)                      # pixel columns, 0.5 mm apart
    xv = np.array([-3e-3, -1e-3, 1e-3, 3e-3])              # foci, 2 mm apart
    s = A.ap_scanline(xi, xv)
    assert s.shape == (1, 17, 1, 1, 4)
    assert np.array_equal(np.nonzero(s[0, :, 0, 0, 1])[0], [6])        # |xi - (-1 mm)| < 0.5 mm: the column at -1 mm only
    assert np.array_equal(np.nonzero(A.ap_scanline(xi, xv, 0.6e-3)[0, :, 0, 0, 1])[0], [5, 6, 7])
    m = A.ap_multiline(xi, xv)
    assert m.shape == (1, 17, 1, 1, 4)
    w = m[0, :, 0, 0, :]                                   # columns x transmits
    inside = (xi >= xv[0]) & (xi <= xv[-1])
    assert np.allclose(w[inside].sum(1), 1.0) and np.all(w[~inside] == 0)    # linear interpolation between the neighbours; nothing outside
    assert np.allclose(w[7], [0, 0.75, 0.25, 0])           # the column at -0.5 mm: 3/4 from the focus at -1 mm, 1/4 from +1 mm
    assert np.allclose(w[6], [0, 1, 0, 0])                 # on a focus: all of it (left == right: the left one takes 1)
    assert A.ap_multiline(xi, xv, xdim=0).shape == (17, 1, 1, 1, 4)
    xn = np.linspace(-3e-3, 3e-3, 13)
    t = A.ap_translating_aperture(xi, xv, xn, [1e-3, 1.5e-3])
    assert t.shape == (1, 17, 1, 13, 4)
    k = 7                                                  # column at -0.5 mm: transmits within 1 mm (the two central foci: no -- only -1 mm ... +0 mm)
    assert np.array_equal(np.nonzero(t[0, k, 0, 6, :])[0], [1])
    assert np.array_equal(np.nonzero(t[0, k, 0, :, 1])[0], np.nonzero(np.abs(xn - xi[k]) <= 1.5e-3)[0])


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["single", "halfT"])
@pytest.mark.parametrize("rule", ["multiline", "scanline"])
def test_reference_shaped_transmit_rules_run_fused(rule, prec):
    """the reference's transmit-side rules come singleton over depth (``1 x I2 x 1 x 1 x M``, src/UltrasoundSystem.m:5071): multiplied out per
    plan, fused kernel with the transmit as the stage element; the translating aperture (pixel x receiver x transmit) is split into its two factors by the host"""
    from qups_amd import apodization as A
    from tests.test_gpu_parity import run_das, run_oracle
    case = make_case(seq="FC", interp="cubic", seed=61, N=32, M=12, I1=140, I2=41, xspan=5e-3)
    xi = np.linspace(-2.5e-3, 2.5e-3, 41)
    xv = np.asarray(case["Pv"])[0]
    x = case["x"]
    q = (lambda a: a.astype(np.float16).astype(np.float64)) if prec == "halfT" else (lambda a: a.astype(np.float32).astype(np.float64))
    if prec == "halfT":
        x = x.real.astype(np.float16).astype(np.float64) + 1j * x.imag.astype(np.float16).astype(np.float64)
    a = q(A.ap_multiline(xi, xv) if rule == "multiline" else A.ap_scanline(xi, xv, 0.3e-3))
    assert a.shape == (1, 41, 1, 1, 12) and a.any()
    ref = run_oracle(case, apod=(a,), x=x)
    out, plan = run_das(case, kernel=2, prec=prec, apod=(a,))
    assert plan.kernel == "tiled" and "roles swapped" in plan.kernel_name()
    assert rel_err(out, ref) <= (2e-5 if prec == "single" else 3e-3)
    if rule == "multiline" and prec == "single":
        t = q(A.ap_translating_aperture(xi, xv, np.asarray(case["Pr"])[0], [0.5e-3, 3e-3]))
        assert t.shape == (1, 41, 1, 32, 12)
        # pixel x receiver x transmit, but an exact product of a transmit-side and a receive-side mask: the host passes the two factors
        # (qups_amd/das_spec.py _split_separable) and the plan runs them fused with per-pair pixel weights
        out, plan = run_das(case, kernel=0, apod=(t,))
        assert plan.kernel == "tiled" and "W=384" in plan.kernel_name() and rel_err(out, run_oracle(case, apod=(t,))) <= 2e-5
        g = t * q(np.random.default_rng(3).uniform(0.5, 1.0, (1, 41, 1, 32, 12)))     # not separable: generic kernel
        out, plan = run_das(case, kernel=0, apod=(g,))
        assert plan.kernel == "generic" and rel_err(out, run_oracle(case, apod=(g,))) <= 1e-4
