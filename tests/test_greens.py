"""Point-scatterer simulator: oracle properties on the CPU; HIP kernel vs oracle and the reference's own
integration criterion (test/BFTest.m:306-316: the beamformed image of greens() data peaks within 1.1 mm of the scatterer) on the GPU."""
from __future__ import annotations

import numpy as np
import pytest

from qups_amd import geometry as G


def _pulse(fc, fsk, ncyc=2.5):
    t = np.arange(-ncyc / fc, ncyc / fc + 0.5 / fsk, 1.0 / fsk)
    return (np.exp(-(t * fc * 1.2) ** 2) * np.exp(2j * np.pi * fc * t)).astype(np.complex64), float(t[0])


def _setup(seed=0, N=8, M=6, I=5, En=1, Em=1, fsr=2.0):
    r = np.random.default_rng(seed)
    fc, c0 = 5e6, 1540.0
    fs = 4 * fc
    Pr = G.linear_array(N, 0.3e-3)[0]
    Pv = G.linear_array(M, 0.4e-3)[0]
    if En > 1:
        Pr = np.stack([Pr + np.array([[dx], [0], [0]]) for dx in np.linspace(-0.1e-3, 0.1e-3, En)], axis=2)
    if Em > 1:
        Pv = np.stack([Pv + np.array([[dx], [0], [0]]) for dx in np.linspace(-0.12e-3, 0.12e-3, Em)], axis=2)
    Ps = np.stack([r.uniform(-3e-3, 3e-3, I), np.zeros(I), r.uniform(5e-3, 12e-3, I)])
    a = (r.uniform(0.5, 1.5, I) * np.exp(2j * np.pi * r.uniform(0, 1, I))).astype(np.complex64)
    x, t0x = _pulse(fc, fsr * fs)
    f32 = lambda v: np.asarray(v, np.float32).astype(np.float64)
    return dict(Ps=f32(Ps), a=a, Pr=f32(Pr), Pv=f32(Pv), x=x, S=700, s0=float(np.float32(4e-6)), t0=float(np.float32(t0x)),
                fs=float(np.float32(fs)), fsr=fsr, cinv=float(np.float32(1 / c0)), R0=float(np.float32(0.3e-3)), c0=c0, fc=fc)


def test_oracle_single_scatterer_arrives_on_time_and_is_linear():
    from oracle import greens_oracle as GO
    g = _setup(I=1)
    y = GO.greens_kernel(g["Ps"], g["a"], g["Pr"], g["Pv"], g["x"], g["S"], g["s0"], g["t0"], g["fs"], g["fsr"], g["cinv"], g["R0"], "cubic")
    assert y.shape == (700, 8, 6)
    for n, m in ((0, 0), (3, 5), (7, 2)):
        tof = (np.linalg.norm(g["Ps"][:, 0] - g["Pr"][:, n]) + np.linalg.norm(g["Ps"][:, 0] - g["Pv"][:, m])) * g["cinv"]
        peak = np.argmax(np.abs(y[:, n, m])) / g["fs"] + g["s0"]
        assert abs(peak - tof) <= 1.0 / g["fs"]                       # the envelope peaks at the two-way time of flight
    g2 = _setup(I=4, seed=3)
    ya = GO.greens_kernel(g2["Ps"], g2["a"], g2["Pr"], g2["Pv"], g2["x"], g2["S"], g2["s0"], g2["t0"], g2["fs"], g2["fsr"], g2["cinv"], g2["R0"], "linear")
    yb = sum(GO.greens_kernel(g2["Ps"][:, [i]], g2["a"][[i]], g2["Pr"], g2["Pv"], g2["x"], g2["S"], g2["s0"], g2["t0"], g2["fs"], g2["fsr"],
                              g2["cinv"], g2["R0"], "linear") for i in range(4))
    assert np.abs(ya - yb).max() <= 1e-12 * np.abs(ya).max()
    # R0 = 0: no propagation loss (the reference's CPU branch)
    y0 = GO.greens_kernel(g["Ps"], g["a"], g["Pr"], g["Pv"], g["x"], g["S"], g["s0"], g["t0"], g["fs"], g["fsr"], g["cinv"], 0.0, "cubic")
    assert abs(np.abs(y0).max() - abs(g["a"][0]) / g["fsr"]) <= 0.02 * abs(g["a"][0]) / g["fsr"]     # unit-peak pulse, no 1/(r1 r2)
    assert np.abs(y).max() > 100 * np.abs(y0).max()                    # with loss: / (r1 r2), r ~ 1e-2 m


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["per-sample", "trains"])
@pytest.mark.parametrize("interp", ["nearest", "linear", "cubic", "lanczos3"])
@pytest.mark.parametrize("prec,En,Em,fsr,R0", [("single", 1, 1, 2.0, None), ("single", 2, 3, 1.0, None), ("double", 1, 2, 4.0, None),
                                                ("single", 1, 1, 2.0, 0.0), ("single", 1, 2, 4.0, None), ("single", 1, 1, 1.5, None)])
def test_greens_kernel_matches_oracle(interp, prec, En, Em, fsr, R0, path, monkeypatch):
    """both kernels of csrc/greens.hip: the per-(entry, sample) one, and -- integer fsr, fp32 -- the impulse trains + one convolution per block
    (QDAS_GREENS_TRAIN_MIN switches per call; fp64 data and fsr = 1.5 stay on the first whatever it says)"""
    import torch
    if path == "trains" and (prec == "double" or fsr != int(fsr)):
        pytest.skip("fp64 data and fractional fsr have one kernel: the per-sample one (covered by the other parameter)")
    monkeypatch.setenv("QDAS_GREENS_TRAIN_MIN", "0" if path == "trains" else "1000000000000")
    from oracle import greens_oracle as GO
    from qups_amd.greens import greens_kernel
    g = _setup(seed=5, N=9, M=7, I=300 if En == 1 else 40, En=En, Em=Em, fsr=fsr)     # > one 256-entry pass
    R0 = g["R0"] if R0 is None else R0
    ref = GO.greens_kernel(g["Ps"], g["a"], g["Pr"], g["Pv"], g["x"], g["S"], g["s0"], g["t0"], g["fs"], fsr, g["cinv"], R0, interp)
    y = greens_kernel(g["Ps"], g["a"], g["Pr"], g["Pv"], g["x"], g["S"], g["s0"], g["t0"], g["fs"], fsr, g["cinv"], R0, interp, prec)
    torch.cuda.synchronize()
    out = y.cpu().numpy()
    assert out.shape == ref.shape
    err = np.abs(out - ref).max() / np.abs(ref).max()
    if interp == "nearest" and prec == "single":
        bad = np.abs(out - ref) / np.abs(ref).max() > 1e-4
        assert bad.mean() < 0.02                                        # an fp32 delay on a rounding boundary picks the neighbour
    else:
        assert err <= (1e-10 if prec == "double" else 3e-4), err


@pytest.mark.gpu
def test_greens_then_das_peaks_at_the_scatterer():
    """the reference's integration criterion (test/BFTest.m:306-316)"""
    import torch
    from qups_amd import das_spec
    from qups_amd.greens import greens
    fc, c0 = 5e6, 1500.0
    fs = 4 * fc
    Pr, nrm = G.linear_array(32, 0.3e-3)
    scat = np.array([[2e-3], [0.0], [15e-3]])                          # test/BFTest.m:28
    wv, t0w = _pulse(fc, 4 * fs)
    y, t0 = greens(Pr, Pr, scat, [1.0], c0, wv, t0w, 4 * fs, fs, R0=c0 / fc, interp="cubic")
    x = np.linspace(-4e-3, 8e-3, 97)
    z = np.linspace(9e-3, 21e-3, 97)
    Pi = G.scan_cartesian(x, z)
    Pv, Nv, opt = G.sequence_args("FSA", tx_pos=Pr, tx_normals=nrm)
    b = das_spec("DAS", Pi, Pr, Pv, Nv, y.contiguous(), t0, fs, c0, *opt, "interp", "cubic")
    torch.cuda.synchronize()
    img = np.abs(b.cpu().numpy())[:, :, 0, 0, 0]
    assert img.max() > 0
    iz, ix = np.unravel_index(np.argmax(img), img.shape)
    assert abs(x[ix] - 2e-3) <= 1.1e-3 and abs(z[iz] - 15e-3) <= 1.1e-3


@pytest.mark.gpu
def test_examples_psf_demo_finds_both_targets():
    """examples/psf_demo.py: greens -> real RF -> hilbert -> DAS with a generated acceptance-angle apodization"""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("psf_demo", os.path.join(os.path.dirname(os.path.dirname(__file__)), "examples", "psf_demo.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    peaks = mod.main()
    (x0, z0, _), (x1, z1, _) = peaks
    assert abs(x0 + 3.0) <= 1.1 and abs(z0 - 22.0) <= 1.1 and abs(x1 - 2.0) <= 1.1 and abs(z1 - 15.0) <= 1.1


@pytest.mark.gpu
def test_greens_impulse_trains_many_scatterers_edges_and_reproducibility(monkeypatch):
    """the impulse-train kernel where it matters and where it could go wrong: thousands of scatterers (coincident slots, several blocks of samples),
    a SHORT waveform whose edges are far from zero (the edge rule per train: a sample whose taps cross the waveform's end is exactly zero, not a
    partial sum), amplitudes over six decades (fixed-point scale), and bit-identical results run after run (integer atomics)"""
    import torch
    from oracle import greens_oracle as GO
    from qups_amd.greens import greens_kernel
    g = _setup(seed=11, N=4, M=3, I=1500, fsr=4.0)
    r = np.random.default_rng(1)
    g["a"] = (g["a"] * 10.0 ** r.uniform(-6, 0, g["a"].shape)).astype(np.complex64)
    x_short = (r.standard_normal(9) + 1j * r.standard_normal(9)).astype(np.complex64)       # 9 waveform samples, no taper
    for x in (g["x"], x_short):
        for interp in ("nearest", "linear", "cubic", "lanczos3"):
            ref = GO.greens_kernel(g["Ps"], g["a"], g["Pr"], g["Pv"], x, 1100, g["s0"], g["t0"], g["fs"], 4.0, g["cinv"], g["R0"], interp)
            monkeypatch.setenv("QDAS_GREENS_TRAIN_MIN", "0")
            y1 = greens_kernel(g["Ps"], g["a"], g["Pr"], g["Pv"], x, 1100, g["s0"], g["t0"], g["fs"], 4.0, g["cinv"], g["R0"], interp, "single")
            y2 = greens_kernel(g["Ps"], g["a"], g["Pr"], g["Pv"], x, 1100, g["s0"], g["t0"], g["fs"], 4.0, g["cinv"], g["R0"], interp, "single")
            monkeypatch.setenv("QDAS_GREENS_TRAIN_MIN", "1000000000000")
            y0 = greens_kernel(g["Ps"], g["a"], g["Pr"], g["Pv"], x, 1100, g["s0"], g["t0"], g["fs"], 4.0, g["cinv"], g["R0"], interp, "single")
            torch.cuda.synchronize()
            assert torch.equal(y1, y2)                                    # order-independent accumulation
            o1, o0 = y1.cpu().numpy(), y0.cpu().numpy()
            den = np.abs(ref).max()
            # (against the float64 oracle both kernels carry the fp32 delay: a sample on a rounding / support boundary picks the neighbour or the zero)
            assert (np.abs(o1 - ref) / den > 1e-4).mean() < (0.02 if interp == "nearest" else 1e-3), (interp, len(x))
            if interp not in ("nearest",) and len(x) > 9:
                assert np.abs(o1 - ref).max() / den <= 3e-4, (interp, len(x))
            assert np.abs(o1 - o0).max() / den <= 2e-5, (interp, len(x))         # the same sum, re-associated
            assert not np.array_equal(o1, o0)                             # (the two kernels round differently: identical bits would mean the switch did nothing)


@pytest.mark.gpu
@pytest.mark.parametrize("interp", ["linear", "lanczos3"])
def test_greens_morton_sorted_scan_changes_no_bit(interp, monkeypatch):
    """from 4096 scatterers on, the impulse-train kernel works on the scatterers in Morton order and visits only the chunks of 256 whose distance
    bounds reach its block of samples (csrc/greens.hip: greens_key_kernel, greens_dist_kernel).  The trains are integer sums, so the image must be
    bit-identical to the unsorted scan (QDAS_GREENS_NO_SORT) -- with coincident scatterers, a degenerate axis (a planar cloud) and non-finite
    positions in the cloud -- and agree with the float64 oracle."""
    import torch
    from oracle import greens_oracle as GO
    from qups_amd.greens import greens_kernel
    g = _setup(seed=5, N=3, M=2, I=4300, fsr=2.0)
    r = np.random.default_rng(2)
    Ps = g["Ps"].copy()
    Ps[1, :] = 0.0                                                        # planar cloud: the y axis of the bounding box has no extent
    Ps[:, 100:140] = Ps[:, 99:100]                                        # 41 coincident scatterers
    g["a"] = (g["a"] * 10.0 ** r.uniform(-3, 0, g["a"].shape)).astype(np.complex64)
    args = (g["a"], g["Pr"], g["Pv"], g["x"], 900, g["s0"], g["t0"], g["fs"], 2.0, g["cinv"], g["R0"], interp)
    monkeypatch.setenv("QDAS_GREENS_TRAIN_MIN", "0")
    ref = GO.greens_kernel(Ps, *args)
    ys = []
    for no_sort in (False, True):
        if no_sort:
            monkeypatch.setenv("QDAS_GREENS_NO_SORT", "1")
        ys.append(greens_kernel(Ps, *args, "single"))
    torch.cuda.synchronize()
    assert torch.equal(ys[0], ys[1])
    o = ys[0].cpu().numpy()
    den = np.abs(ref).max()
    assert den > 0 and np.abs(o - ref).max() / den <= 3e-4
    if interp == "linear":
        # the temporaries from per-call blocks instead of the stream's arena (csrc/scratch.hip: what clouds beyond 64 MiB of tables get): the same bits
        monkeypatch.delenv("QDAS_GREENS_NO_SORT")
        monkeypatch.setenv("QDAS_SCRATCH_ARENA_MAX_MB", "0")
        y_blk = greens_kernel(Ps, *args, "single")
        monkeypatch.delenv("QDAS_SCRATCH_ARENA_MAX_MB")
        torch.cuda.synchronize()
        assert torch.equal(y_blk, ys[0])
        monkeypatch.setenv("QDAS_GREENS_NO_SORT", "1")
        # purely real amplitudes (the usual clouds) take the path that skips the imaginary trains' arithmetic (TrainBlock::retire, a_cplx == false)
        a_re = np.abs(g["a"]).astype(np.complex64)
        args_re = (a_re,) + args[1:]
        monkeypatch.delenv("QDAS_GREENS_NO_SORT")
        y_re = greens_kernel(Ps, *args_re, "single").cpu().numpy()
        ref_re = GO.greens_kernel(Ps, *args_re)
        assert np.abs(y_re - ref_re).max() / np.abs(ref_re).max() <= 3e-4
        monkeypatch.setenv("QDAS_GREENS_NO_SORT", "1")
    # non-finite positions: the scatterer contributes NaN / nothing exactly as in the unsorted scan
    monkeypatch.delenv("QDAS_GREENS_NO_SORT")
    Pn = Ps.copy(); Pn[0, 7] = np.nan; Pn[2, 4200] = np.inf
    y_s = greens_kernel(Pn, *args, "single")
    monkeypatch.setenv("QDAS_GREENS_NO_SORT", "1")
    y_u = greens_kernel(Pn, *args, "single")
    torch.cuda.synchronize()
    assert torch.equal(torch.view_as_real(y_s).view(torch.int32), torch.view_as_real(y_u).view(torch.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("QDAS_GREENS_FUZZ", "16"))))
def test_greens_random_configuration(seed, monkeypatch):
    """random simulator configurations (elements, sub-apertures, scatterer counts around the chunk sizes of both kernels, record lengths around the block
    sizes, integer and fractional fsr, every interpolator, with and without propagation loss, waveforms from 3 samples up) through BOTH kernels against
    the float64 oracle; where both apply they agree to 5e-6 of the peak"""
    import torch
    from oracle import greens_oracle as GO
    from qups_amd.greens import greens_kernel
    r = np.random.default_rng(1000 + seed)
    N, M = int(r.integers(1, 7)), int(r.integers(1, 6))
    En, Em = int(r.choice([1, 1, 2, 3])), int(r.choice([1, 1, 2]))
    fsr = float(r.choice([1.0, 2.0, 3.0, 4.0, 8.0, 1.5, 2.5]))
    I = int(r.choice([1, 3, 255, 257, 1023, 1025, 2500]))
    S = int(r.choice([1, 63, 64, 65, 511, 512, 513, 700, 1300]))
    interp = str(r.choice(["nearest", "linear", "cubic", "lanczos3"]))
    g = _setup(seed=seed, N=N, M=M, I=I, En=En, Em=Em, fsr=fsr)
    R0 = 0.0 if r.random() < 0.25 else g["R0"]
    x = g["x"]
    if r.random() < 0.3:
        Tn = int(r.integers(3, 12))
        x = (r.standard_normal(Tn) + 1j * r.standard_normal(Tn)).astype(np.complex64)
    ref = GO.greens_kernel(g["Ps"], g["a"], g["Pr"], g["Pv"], x, S, g["s0"], g["t0"], g["fs"], fsr, g["cinv"], R0, interp)
    out = {}
    for path, v in (("trains", "0"), ("per-sample", "1000000000000")):
        monkeypatch.setenv("QDAS_GREENS_TRAIN_MIN", v)
        out[path] = greens_kernel(g["Ps"], g["a"], g["Pr"], g["Pv"], x, S, g["s0"], g["t0"], g["fs"], fsr, g["cinv"], R0, interp, "single").cpu().numpy()
    den = np.abs(ref).max()
    if den == 0:
        assert not out["trains"].any() and not out["per-sample"].any()
        return
    for path, o in out.items():
        assert o.shape == ref.shape
        bad = np.abs(o - ref) / den > 3e-4               # (an fp32 delay on a rounding / support boundary picks the neighbouring sample or the zero)
        assert bad.mean() < (0.03 if interp == "nearest" or len(x) < 16 else 1e-3), (path, interp, float(bad.mean()))
    # the same sum, re-associated (the per-sample kernel adds thousands of fp32 terms in sequence; the trains are exact sums rounded once).  Where the
    # interpolant itself jumps -- nearest at u = 1/2, any interpolator at the end of a short untapered waveform -- the kernels' tap indices (floor(fsr (s - d))
    # against Q s - ceil(Q d), both from the same fp32 d) may land on either side for an entry on the boundary: a few samples, as against the oracle
    dd = np.abs(out["trains"] - out["per-sample"]) / den
    if interp == "nearest" or len(x) < 16:
        assert (dd > 5e-5).mean() < 0.03, float((dd > 5e-5).mean())
    else:
        assert dd.max() <= 5e-5, float(dd.max())
