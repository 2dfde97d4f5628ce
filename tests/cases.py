"""Seeded problem builders shared by the CPU and GPU tests (inputs are float32-exact so that the
float64 oracle and the fp32 kernels see IDENTICAL inputs)."""
from __future__ import annotations

import numpy as np

from qups_amd import geometry as G


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def make_case(seq="FSA", N=16, M=None, I1=48, I2=12, T=None, interp="cubic", seed=0, fc=5e6, c0=1540.0,
              data="targets", t0=None, convex=False, pitch=0.3e-3, zlim=(4e-3, 14e-3), xspan=6e-3, noise=0.01):
    """A small imaging problem: linear (or convex) array, Cartesian scan, FSA / PW / FC / DV sequence.

    Returns a dict with das_spec positional args (Pi, Pr, Pv, Nv, x, t0, fs, c) + option strings and
    the matching keyword args for the oracle.
    """
    rng = np.random.default_rng(seed)
    fs = 4 * fc
    lam = c0 / fc
    if convex:
        Pr, nrm = G.convex_array(N, 40e-3, 0.6)
    else:
        Pr, nrm = G.linear_array(N, pitch)
    z = np.linspace(zlim[0], zlim[1], I1)
    xx = np.linspace(-xspan / 2, xspan / 2, I2)
    Pi = G.scan_cartesian(xx, z)
    if seq == "FSA":
        Pv, Nv, opt = G.sequence_args("FSA", tx_pos=Pr, tx_normals=nrm)
        M = N
    elif seq == "PW":
        M = M or 7
        th = np.deg2rad(np.linspace(-15, 15, M))
        Pv, Nv, opt = G.sequence_args("PW", focus=np.stack([np.sin(th), 0 * th, np.cos(th)]))
    elif seq in ("FC", "DV"):
        M = M or 6
        xf = np.linspace(-2e-3, 2e-3, M)
        zf = 9e-3 if seq == "FC" else -6e-3
        Pv, Nv, opt = G.sequence_args(seq, focus=np.stack([xf, 0 * xf, 0 * xf + zf]))
    else:
        raise ValueError(seq)
    VS = "plane-waves" not in opt
    DV = "diverging-waves" in opt
    Pi, Pr, Pv, Nv = f32(Pi), f32(Pr), f32(Pv), f32(Nv)
    # record length: cover the longest two-way path (+ margin) unless T is forced
    far = np.array([[xx[-1]], [0.0], [z[-1]]])
    dmax = 2.2 * np.linalg.norm(far - np.array([[Pr[0].min()], [0.0], [Pr[2].min()]]))
    if t0 is None:
        t0 = float(np.float32(-8 / fs)) if seq != "FC" else float(np.float32(-(abs(zf) / c0) - 8 / fs))
    if T is None:
        T = int(np.ceil((dmax / c0 - (t0 if np.isscalar(t0) else np.min(t0))) * fs)) + 16
    if data == "targets":
        sc = np.array([[0.0, 1.2e-3, -1.5e-3], [0, 0, 0], [np.mean(zlim), zlim[0] + 2e-3, zlim[1] - 2e-3]])
        x = G.point_target_data(sc, [1.0, 0.7, 0.5], Pr, Pv, Nv, VS=VS, DV=DV, c0=c0, fs=fs, fc=fc, T=T, t0=t0,
                                dtype=np.complex128)
        x = x + noise * (rng.standard_normal(x.shape) + 1j * rng.standard_normal(x.shape))
    else:
        x = rng.standard_normal((T, N, M)) + 1j * rng.standard_normal((T, N, M))
    x = x.astype(np.complex64)
    return dict(Pi=Pi, Pr=Pr, Pv=Pv, Nv=Nv, x=x, t0=t0, fs=float(np.float32(fs)), c=float(np.float32(c0)),
                opt=list(opt), VS=VS, DV=DV, interp=interp, N=N, M=M, T=T, fc=fc)


def cinv_f32(c):
    """the value the device path actually uses: float32(1/c) (reference kern/das_spec.m:170,244)"""
    return 1.0 / np.float64(np.float32(1.0 / np.float64(c)))


def rel_err(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    den = np.abs(b).max()
    return float(np.abs(a - b).max() / (den if den > 0 else 1.0))


def stock_kernel(name: str) -> bool:
    """the plan runs an instantiation of the fused kernel with run-time sizes -- carried by libqdas.so ("[prebuilt]") or built on demand from
    the same template arguments ("[built on demand <key>]", csrc/das_tile_cfg.h TILE_PREBUILT) -- not a plan-specialised hiprtc build ("[jit <key>]")"""
    return "[prebuilt]" in name or "[built on demand " in name
