"""Edge cases of the DAS path on the GPU: empty problems, 3-D scans over a matrix array, non-finite geometry, very short
records, apertures too large for the LDS header, frames = 0 -- through the C ABI (ctypes) like every GPU test."""
from __future__ import annotations

import ctypes as C

import numpy as np
import pytest

from tests.cases import cinv_f32, make_case, rel_err

pytestmark = pytest.mark.gpu


def _oracle(case, **kw):
    from oracle import das_oracle as O
    return O.das_spec("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], case["x"], case["t0"], case["fs"], cinv_f32(case["c"]),
                      VS=case["VS"], DV=case["DV"], interp=case["interp"], **kw)


def _run(case, kernel=0, **kw):
    import torch
    from qups_amd import das_spec
    y, plan = das_spec("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], torch.from_numpy(case["x"]), case["t0"], case["fs"],
                       case["c"], *case["opt"], "interp", case["interp"], return_plan=True, kernel=kernel, **kw)
    torch.cuda.synchronize()
    return y.cpu().numpy(), plan


def test_empty_problems_through_the_c_abi():
    """I = 0 writes nothing; N = 0 or M = 0 or T = 0 (an empty sum) zero-fills; F = 0 is a no-op"""
    import torch
    from qups_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
    Pi, Pr, Pv, Nv, cinv = z(3, 6), z(3, 4), z(4, 3), z(3, 3), torch.full((1,), 1 / 1540, dtype=torch.float32, device=dev)
    acs = (C.c_uint64 * 6)(0, 0, 0, 0, 0, 0)
    for (T, N, M, I1) in ((16, 4, 3, 0), (16, 0, 3, 6), (16, 4, 0, 6), (0, 4, 3, 6)):
        d = _lib.Desc()
        d.sz = _lib.Sizes(T, N, M, I1, 1, 1, 0, 1, 1, 1, _lib.QDAS_F32)
        d.fs = 20e6
        d.Pi, d.Pr, d.Pv, d.Nv, d.cinv = (C.c_void_p(t.data_ptr()) for t in (Pi, Pr, Pv, Nv, cinv))
        d.acstride, d.mem, d.device = acs, _lib.MEM_DEVICE, 0
        h = C.c_void_p()
        _lib.check(L.qdas_plan_create(C.byref(h), C.byref(d)))
        x = torch.ones((max(T * N * M, 1), 2), dtype=torch.float32, device=dev)
        y = torch.full((max(I1, 1), 2), 7.0, dtype=torch.float32, device=dev)
        _lib.check(L.qdas_plan_execute(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), None))
        _lib.check(L.qdas_plan_execute_frames(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), 0, 0, 0, None))
        torch.cuda.synchronize()
        assert float(y.abs().max()) == (7.0 if I1 == 0 else 0.0), (T, N, M, I1)
        L.qdas_plan_destroy(h)


@pytest.mark.parametrize("kernel", [1, 2])
def test_volume_scan_over_a_matrix_array(kernel):
    """I3 > 1 and elements / pixels with y != 0 (matrix probe, 3-D scan)"""
    import torch
    from qups_amd import das_spec
    from oracle import das_oracle as O
    r = np.random.default_rng(7)
    fc, c0 = 5e6, 1540.0
    fs = 4 * fc
    ex, ey = np.meshgrid((np.arange(4) - 1.5) * 0.3e-3, (np.arange(4) - 1.5) * 0.3e-3, indexing="ij")
    Pr = np.stack([ex.ravel(), ey.ravel(), 0 * ex.ravel()])
    z = 5e-3 + np.arange(70) * 0.08e-3
    xs = np.linspace(-1e-3, 1e-3, 9)
    ys = np.linspace(-0.8e-3, 0.8e-3, 3)
    Z, X, Y = np.meshgrid(z, xs, ys, indexing="ij")
    Pi = np.stack([X, Y, Z])                                            # 3 x I1 x I2 x I3
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    Pi, Pr = f32(Pi), f32(Pr)
    T = 420
    x = (r.standard_normal((T, 16, 16)) + 1j * r.standard_normal((T, 16, 16))).astype(np.complex64)
    Nv = np.tile(np.array([[0.0], [0.0], [1.0]]), (1, 16))
    ref = O.das_spec("DAS", Pi, Pr, Pr, Nv, x, 0.0, float(np.float32(fs)), cinv_f32(float(np.float32(c0))), VS=True, DV=True, interp="cubic")
    y, plan = das_spec("DAS", Pi, Pr, Pr, Nv, torch.from_numpy(x), 0.0, float(np.float32(fs)), float(np.float32(c0)),
                       "virtual-source", "diverging-waves", "interp", "cubic", return_plan=True, kernel=kernel)
    torch.cuda.synchronize()
    assert tuple(y.shape)[:3] == (70, 9, 3) and np.abs(ref).max() > 0
    assert rel_err(y.cpu().numpy(), ref) <= (1e-4 if kernel == 1 else 3e-5)


@pytest.mark.parametrize("kernel", [1, 2])
def test_non_finite_pixels_give_exact_zeros(kernel):
    """NaN / Inf pixel coordinates: that pixel is exactly 0 (non-finite delay -> out of support), its neighbours are untouched"""
    case = make_case(seq="FSA", interp="linear", seed=51, N=16, I1=130, I2=18, zlim=(4e-3, 17e-3), xspan=3e-3)
    ref = _oracle(case)
    bad = [(5, 2), (64, 0), (129, 17), (77, 9)]
    case["Pi"] = case["Pi"].copy()
    for k, (a, b) in enumerate(bad):
        case["Pi"][k % 3, a, b, 0] = np.nan if k % 2 == 0 else np.inf
    out, plan = _run(case, kernel=kernel)
    assert np.all(np.isfinite(out))
    mask = np.zeros(out.shape[:2], bool)
    for a, b in bad:
        assert out[a, b].ravel()[0] == 0
        mask[a, b] = True
    assert rel_err(out[~mask], ref[~mask]) <= 1e-4


def test_very_short_record_and_huge_aperture_route_to_the_generic_kernel():
    from qups_amd import _lib
    case = make_case(seq="PW", interp="cubic", seed=52, N=5, M=3, I1=20, I2=4, T=6, data="noise", zlim=(0.1e-3, 0.4e-3), t0=0.0)
    out, plan = _run(case)
    assert plan.kernel == "generic"                                     # T < 8: nothing to stage
    assert rel_err(out, _oracle(case)) <= 1e-4 or np.abs(_oracle(case)).max() == 0
    with pytest.raises(_lib.QdasError, match="T >= 8"):
        _run(case, kernel=2)
    big = make_case(seq="PW", interp="linear", seed=53, N=4000, M=2, I1=8, I2=2, data="noise", pitch=0.01e-3)
    out, plan = _run(big)
    assert plan.kernel == "generic"                                     # N + M receivers / transmits do not fit the LDS header
    assert rel_err(out, _oracle(big)) <= 1e-4
    with pytest.raises(_lib.QdasError, match="LDS header"):
        _run(big, kernel=2)


@pytest.mark.parametrize("F", [1, 4])
@pytest.mark.parametrize("seq", ["FSA", "PW"])
def test_host_resident_inputs_through_the_c_abi(F, seq):
    """QDAS_MEM_HOST (what the MEX shim passes): geometry, data and output are HOST arrays; the plan stages them -- frame
    sequences double-buffered on a copy stream -- and must give what the device-resident path gives"""
    import torch
    from qups_amd import DasPlan, _lib, build_problem, parse_options
    from qups_amd.das_spec import _cast_data, _colmajor
    case = make_case(seq=seq, interp="cubic", seed=91, N=16, I1=140, I2=18, zlim=(4e-3, 15e-3), xspan=3e-3, data="noise")
    r = np.random.default_rng(92)
    xs = np.stack([case["x"]] + [(r.standard_normal(case["x"].shape) + 1j * r.standard_normal(case["x"].shape)).astype(np.complex64)
                                 for _ in range(F - 1)], axis=3)
    xt = torch.from_numpy(np.ascontiguousarray(xs))
    prob = build_problem("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], tuple(xt.shape), case["t0"], case["fs"], case["c"],
                         parse_options(xt, list(case["opt"]) + ["interp", "cubic"]))
    # device-resident reference run
    plan = DasPlan(prob, kernel=2)
    yd = plan.execute_colmajor(_colmajor(_cast_data(xt, prob.prec, plan.device)), F)
    torch.cuda.synchronize()
    yd = yd.cpu().numpy().reshape(F, -1)
    # host-resident run through ctypes
    L = _lib.lib()
    acs = (C.c_uint64 * len(prob.acstride))(*[int(v) for v in prob.acstride])
    d = _lib.Desc()
    d.sz = _lib.Sizes(prob.T, prob.N, prob.M, prob.Isz[0], prob.Isz[1], prob.Isz[2], prob.S, prob.flag, int(prob.VS), int(prob.DV), _lib.QDAS_F32)
    d.fs, d.fmod = prob.fs, prob.fmod
    keep = [np.ascontiguousarray(a) for a in (prob.Pi, prob.Pr, prob.Pv, prob.Nv, prob.cinv)]
    d.Pi, d.Pr, d.Pv, d.Nv, d.cinv = (C.c_void_p(a.ctypes.data) for a in keep)
    d.acstride, d.mem, d.device, d.kernel = acs, _lib.MEM_HOST, 0, _lib.KERNEL_TILED
    h = C.c_void_p()
    _lib.check(L.qdas_plan_create(C.byref(h), C.byref(d)))
    xh = np.ascontiguousarray(xs.transpose(3, 2, 1, 0))               # (F, M, N, T): MATLAB memory order of T x N x M x F
    I = prob.I
    yh = np.full((F, I), 7 + 7j, np.complex64)
    _lib.check(L.qdas_plan_execute_frames(h, C.c_void_p(xh.ctypes.data), C.c_void_p(yh.ctypes.data), F, prob.T * prob.N * prob.M, I, None))
    L.qdas_plan_destroy(h)
    assert np.abs(yh - yd).max() / np.abs(yd).max() <= 3e-5          # (frame pairs on the device path: another summation order)


@pytest.mark.parametrize("kernel,tpose", [(2, False), (1, False), (2, True)])
def test_channel_data_larger_than_4_GiB(kernel, tpose, monkeypatch):
    """64-bit addressing: a 4.4 GB acquisition whose only non-zero traces lie behind the 4 GiB mark must beamform exactly like
    the small acquisition made of those transmits alone (both kernels; the tiled kernel's DMA descriptors are per transmit block).
    Transposed data of more than 2 GiB has receiver strides beyond the 32-bit DMA offsets of one descriptor: the plan picks the
    re-basing instantiation of the general kernel (launch configuration 9)."""
    import torch
    from qups_amd import das_spec
    from qups_amd import geometry as G
    monkeypatch.setenv("QDAS_NO_WIDE", "1")                              # (likewise: the twin alone would get the 384-sample windows the re-basing configuration has not)
    monkeypatch.setenv("QDAS_NO_ROLE_SWAP", "1")                          # (the 8-transmit twin would otherwise run with the apertures' roles swapped: another summation order)
    T, N, M, Ml = 4096, 256, 520, 8                                       # 4096 * 256 * 520 * 8 B = 4.36 GB
    fc, c0 = 5e6, 1540.0
    fs = 4 * fc
    Pr, nrm = G.linear_array(N, 0.2e-3)
    th = np.deg2rad(np.linspace(-20, 20, M))
    Pv, Nv, opt = G.sequence_args("PW", focus=np.stack([np.sin(th), 0 * th, np.cos(th)]))
    Pi = G.scan_cartesian(np.linspace(-5e-3, 5e-3, 48), np.linspace(5e-3, 30e-3, 160))
    g = torch.Generator(device="cuda").manual_seed(5)
    tail = torch.view_as_complex(torch.randn((T, N, Ml, 2), generator=g, device="cuda", dtype=torch.float32))
    x = torch.zeros((T, N, M), dtype=torch.complex64, device="cuda")
    x[:, :, M - Ml:] = tail
    f32 = lambda a: np.asarray(a, np.float32)
    xa, ta = (x.permute(0, 2, 1).contiguous(), tail.permute(0, 2, 1).contiguous()) if tpose else (x, tail)
    big, plan = das_spec("DAS", f32(Pi), f32(Pr), f32(Pv), f32(Nv), xa, -2e-6, fs, c0, *opt, "interp", "cubic", "transpose", tpose,
                         return_plan=True, kernel=kernel)
    last = lambda P: f32(P) if np.asarray(P).shape[1] == 1 else f32(P)[:, M - Ml:]
    small = das_spec("DAS", f32(Pi), f32(Pr), last(Pv), last(Nv), ta, -2e-6, fs, c0, *opt, "interp", "cubic",
                     "transpose", tpose, kernel=kernel or 1)          # (the same kernel on both sides: same fp32 delay rounding)
    torch.cuda.synchronize()
    assert plan.kernel == ("tiled" if kernel == 2 else "generic")
    if ",big" in plan.kernel_name():                                      # the re-basing instantiation never runs in lateral-mirror mode (ADVICE r3:
        assert not plan.mirror, plan.kernel_name()                        # the mode was decided before the stride check and every execute failed)
    b, s = big.cpu().numpy(), small.cpu().numpy()
    assert np.abs(s).max() > 0 and rel_err(b, s) <= 2e-6


@pytest.mark.parametrize("tpose", [False, True])
def test_reciprocal_mode_beyond_2_GiB(tpose, monkeypatch):
    """a 2.6 GB full-synthetic-aperture acquisition: the mirror traces of the reciprocal mode walk the whole frame (descriptor
    re-basing); the result must equal the general tiled kernel's"""
    import torch
    from qups_amd import das_spec
    from qups_amd import geometry as G
    T, N = 3072, 320                                                      # 3072 * 320 * 320 * 8 B = 2.5 GB
    fc, c0 = 5e6, 1540.0
    fs = 4 * fc
    Pr, nrm = G.linear_array(N, 0.2e-3)
    Pv, Nv, opt = G.sequence_args("FSA", tx_pos=Pr, tx_normals=nrm)
    Pi = G.scan_cartesian(np.linspace(-6e-3, 6e-3, 64), np.linspace(5e-3, 5e-3 + 127 * 77e-6, 128))
    g = torch.Generator(device="cuda").manual_seed(6)
    x = torch.view_as_complex(torch.randn((T, N, N, 2), generator=g, device="cuda", dtype=torch.float32))
    f32 = lambda a: np.asarray(a, np.float32)
    ys, plan = das_spec("DAS", f32(Pi), f32(Pr), f32(Pv), f32(Nv), x, 0.0, fs, c0, *opt, "interp", "lanczos3", "transpose", tpose, return_plan=True, kernel=2)
    monkeypatch.setenv("QDAS_NO_SYM", "1")
    if tpose:                                                             # general tiled kernel: not for transposed data of this size
        x = x.permute(0, 2, 1).contiguous()
    yg = das_spec("DAS", f32(Pi), f32(Pr), f32(Pv), f32(Nv), x, 0.0, fs, c0, *opt, "interp", "lanczos3", kernel=2)
    torch.cuda.synchronize()
    assert plan.kernel == "tiled" and plan.reciprocal
    assert rel_err(ys.cpu().numpy(), yg.cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("shape", [(70, 3, 130), (1, 5, 64), (257, 1, 33), (64, 2, 1), (5, 7)])
@pytest.mark.parametrize("dtype", ["float16", "float32", "complex64", "complex128"])
def test_colmajor_layout_kernel(shape, dtype):
    """qdas_permute3 (row-major host arrays -> the column-major order of the ABI): equals torch's permute, ragged tiles, 2-16 byte elements"""
    import torch
    from qups_amd.das_spec import _colmajor
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    t = torch.randn(shape + ((2,) if dtype.startswith("complex") else ()), generator=g, device="cuda",
                    dtype=torch.float64 if dtype == "complex128" else torch.float32)
    t = torch.view_as_complex(t) if dtype.startswith("complex") else t.to(getattr(torch, dtype))
    t = t.contiguous()
    out = _colmajor(t)
    ref = t.permute(*reversed(range(t.ndim))).contiguous()
    assert out.shape == ref.shape and out.is_contiguous() and torch.equal(torch.view_as_real(out) if out.is_complex() else out,
                                                                            torch.view_as_real(ref) if ref.is_complex() else ref)


@pytest.mark.parametrize("seq,M", [("FC", 1), ("FC", 2), ("DV", 1)])
def test_roles_swapped_onto_one_or_two_transmits_is_reproducible(seq, M, monkeypatch):
    """Round 6, fuzz seed 126301 with hiprtc builds forced: 48 receivers, ONE focused (or diverging) transmit, a coarse pixel grid -- the plan swaps the roles
    of the apertures (the transmit becomes the stage side; 16-element stages of 384-sample windows).  The plan-specialised build of that shape SPILLED
    352 registers, and a register spilled between an inline-asm LDS read and the hand-placed wait for it is stored before its data has arrived: images
    that differed from run to run in one wave of a tile.  A hiprtc build that uses scratch memory is no longer used (``csrc/qdas_api.hip``
    ``jit_get_kernel_nospill``).  Here: noise frames (smooth targets hide a wrong sample), six runs bit for bit the same, against the oracle."""
    import torch
    from oracle import das_oracle as O
    from qups_amd import DasPlan, build_problem, parse_options
    from qups_amd.das_spec import _cast_data, _colmajor
    monkeypatch.setenv("QDAS_TILE_Z", "8")
    I1, I2, coarse = 184, 24, 5
    case = make_case(seq=seq, interp="cubic", seed=126301, N=48, M=M, I1=I1, I2=I2, zlim=(4e-3, 4e-3 + I1 * 0.1e-3 * coarse), xspan=2e-3 * coarse, data="noise")
    xs = np.swapaxes(case["x"], 1, 2)[..., None]
    xt = torch.from_numpy(np.ascontiguousarray(xs))
    opts = list(case["opt"]) + ["interp", "cubic", "input-precision", "single", "transpose", True]
    prob = build_problem("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], tuple(xt.shape), case["t0"], case["fs"], case["c"], parse_options(xt, opts))
    ref = O.das_spec("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], case["x"], case["t0"], case["fs"], cinv_f32(case["c"]), VS=case["VS"], DV=case["DV"],
                     interp="cubic", apod=(), fmod=0.0).reshape(-1, order="F")
    for jit in (True, False):
        with DasPlan(prob, kernel=2, jit=jit) as plan:
            assert "roles swapped" in plan.kernel_name(), plan.kernel_name()
            xc = _colmajor(_cast_data(xt, prob.prec, plan.device))
            ys = [plan.execute_colmajor(xc, 1).clone() for _ in range(6)]
            torch.cuda.synchronize()
            for y in ys[1:]:
                assert torch.equal(y, ys[0]), (jit, plan.kernel_name())
            out = ys[0].to(torch.complex64).cpu().numpy().reshape(-1)
            tol = 1e-4 * (2 * coarse if plan.fallback_tiles() else 1)
            assert rel_err(out, ref) <= tol, (jit, plan.kernel_name(), rel_err(out, ref))
