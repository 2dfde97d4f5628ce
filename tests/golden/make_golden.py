#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ (run from the repo root):

    python tests/golden/make_golden.py

The reference (MATLAB + CUDA) cannot run in this environment and ships no stored vectors for this path
(SURVEY.md section 4), so the fixtures are produced by the float64 oracle (oracle/das_oracle.py) on
small, seeded, closed-form inputs.  They are DATA (inputs + expected outputs), used to
  * freeze the oracle (tests/test_oracle_pins.py): any later edit of the oracle that changes numbers fails;
  * check the C oracle and the HIP kernels against values that do not depend on the code under test.

F1  interpTest fixture of the reference (test/interpTest.m:33-43): [I,T,N,M,F] = [16,32,4,3,2],
    x0 = exp(2j pi (1/2 + f/2 n/4) t), t = (0:T-1)/T, tau = t1 + t2 -> sampled with the 4 interpolators.
F2  PSF: 32-element array, FSA / PW / FC, 64 x 48 lambda/8 grid around (2, 0, 15) mm, analytic
    Gaussian-pulse echoes (test/BFTest.m:28-29,92-97 geometry), cubic -> full image + argmax.
F3  modes: DAS / SYN / MUL / BF on a tiny problem (identity sum(BF) == DAS checked in the tests).
F4  edge cases of the interpolators (tau around 0 and T-1).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import das_oracle as O  # noqa: E402
from qups_amd import geometry as G  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def f1():
    I, T, N, M, F = 16, 32, 4, 3, 2
    i = np.arange(I)[:, None, None]
    n = np.arange(N)[None, :, None]
    m = np.arange(M)[None, None, :]
    t1 = 4 + (T - 8) * ((1 + i) / I * 1 / N * (1 + m) / M)      # I x 1 x M
    t2 = 4 + (T - 8) * 1 / I * (1 + n) / N                      # 1 x N x 1
    tau = t1 + t2                                               # I x N x M
    t = (np.arange(T) / T)[:, None, None]
    f = np.arange(F)[None, None, :]
    nn = np.arange(N)[None, :, None]
    x0 = np.exp(2j * np.pi * (0.5 + f / 2 * nn / 4) * t)        # T x N x F
    out = {"tau": tau, "x0": x0}
    for terp in ("nearest", "linear", "cubic", "lanczos3", "cubic_dev"):
        y = np.zeros((I, N, M, F), np.complex128)
        for fi in range(F):
            for mi in range(M):
                y[:, :, mi, fi] = O.sample(x0[:, :, fi], tau[:, :, mi], terp)
        out["y_" + terp] = y
    np.savez_compressed(os.path.join(OUT, "f1_interptest.npz"), **out)


def psf_case(seq):
    c0, fc = 1500.0, 6e6
    fs = 4 * fc
    lam = c0 / fc
    N = 32
    Pr, nrm = G.linear_array(N, 0.2e-3)
    sc = np.array([[2e-3], [0.0], [15e-3]])
    dz = (lam / 8) * (np.arange(64) - 32)
    dx = (lam / 8) * (np.arange(48) - 24)
    Pi = G.scan_cartesian(sc[0, 0] + dx, sc[2, 0] + dz)
    if seq == "FSA":
        Pv, Nv, opt = G.sequence_args("FSA", tx_pos=Pr, tx_normals=nrm)
    elif seq == "PW":
        th = np.deg2rad(np.arange(-10, 11, 5.0))
        Pv, Nv, opt = G.sequence_args("PW", focus=np.stack([np.sin(th), 0 * th, np.cos(th)]))
    else:
        xf = np.linspace(-3e-3, 3e-3, 8) + 2e-3
        Pv, Nv, opt = G.sequence_args("FC", focus=np.stack([xf, 0 * xf, 0 * xf + 50e-3]))
    VS, DV = "plane-waves" not in opt, "diverging-waves" in opt
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    Pi, Pr_, Pv, Nv = f32(Pi), f32(Pr), f32(Pv), f32(Nv)
    t0 = float(np.float32(-50e-3 / c0 - 2e-6)) if seq == "FC" else float(np.float32(-1e-6))
    T = int(np.ceil((2 * 22e-3 / c0 - t0 + (50e-3 / c0 if seq == "FC" else 0)) * fs)) + 32
    x = G.point_target_data(sc, [1.0], Pr_, Pv, Nv, VS=VS, DV=DV, c0=c0, fs=fs, fc=fc, T=T, t0=t0, dtype=np.complex64)
    c_eff = 1.0 / np.float64(np.float32(1.0 / c0))
    y = O.das_spec("DAS", Pi, Pr_, Pv, Nv, x, t0, float(np.float32(fs)), c_eff, VS=VS, DV=DV, interp="cubic")
    return dict(Pi=Pi, Pr=Pr_, Pv=Pv, Nv=Nv, x=x, t0=t0, fs=float(np.float32(fs)), c=c0, VS=VS, DV=DV, y=y[..., 0, 0],
                scat=sc[:, 0], dx=dx, dz=dz)


def f2():
    out = {}
    for seq in ("FSA", "PW", "FC"):
        for k, v in psf_case(seq).items():
            out[f"{seq}_{k}"] = v
    np.savez_compressed(os.path.join(OUT, "f2_psf.npz"), **out)


def f3():
    rng = np.random.default_rng(7)
    I1, I2, N, M, T = 10, 4, 5, 3, 200
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    Pi = f32(G.scan_cartesian(np.linspace(-2e-3, 2e-3, I2), np.linspace(3e-3, 6e-3, I1)))
    Pr = f32(G.linear_array(N, 0.5e-3)[0])
    th = np.deg2rad(np.linspace(-8, 8, M))
    Pv, Nv = np.zeros((3, 1)), f32(np.stack([np.sin(th), 0 * th, np.cos(th)]))
    x = (rng.standard_normal((T, N, M)) + 1j * rng.standard_normal((T, N, M))).astype(np.complex64)
    a1 = f32(rng.uniform(0.2, 1, (I1, 1, 1, N, 1)))
    a2 = f32(rng.uniform(0.2, 1, (1, I2, 1, 1, M)))
    fs, c, t0 = 20e6, 1.0 / np.float64(np.float32(1 / 1540.0)), float(np.float32(-2e-7))
    out = dict(Pi=Pi, Pr=Pr, Pv=Pv, Nv=Nv, x=x, a1=a1, a2=a2, fs=fs, c=1540.0, t0=t0)
    for fun in ("DAS", "SYN", "MUL", "BF"):
        out["y_" + fun] = O.das_spec(fun, Pi, Pr, Pv, Nv, x, t0, fs, c, VS=False, interp="linear", apod=(a1, a2))
    out["y_DAS_fmod"] = O.das_spec("DAS", Pi, Pr, Pv, Nv, x, t0, fs, c, VS=False, interp="linear", fmod=float(np.float32(3e6)))
    np.savez_compressed(os.path.join(OUT, "f3_modes.npz"), **out)


def f4():
    rng = np.random.default_rng(11)
    T = 12
    x = (rng.standard_normal(T) + 1j * rng.standard_normal(T))
    s = np.array([-1.0, -0.5, -0.25, -1e-9, 0.0, 0.25, 0.5, 0.75, 1.0, 1.5, 2.0, 5.5, T - 3.0, T - 2.5, T - 2.0, T - 1.5, T - 1.0,
                  T - 0.75, T - 0.5, T - 0.25, T, T + 3.0, np.inf, -np.inf, np.nan])
    out = {"x": x, "s": s}
    for terp in ("nearest", "linear", "cubic", "lanczos3"):
        out["y_" + terp] = O.sample(x[:, None], s[:, None], terp)[:, 0]
    np.savez_compressed(os.path.join(OUT, "f4_edges.npz"), **out)


if __name__ == "__main__":
    f1(); f2(); f3(); f4()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")
