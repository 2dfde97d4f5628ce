"""Geometry producers consumed by the DAS path -- host logic, numpy only.

The reference keeps these in its definition classes; only the few formulas whose OUTPUT feeds the
kernel are restated here (``3 x N`` element positions / normals, ``3 x I1 x I2 x I3`` pixel grids,
and the per-sequence ``(Pv, Nv, options)`` mapping of ``UltrasoundSystem.DAS``).
"""
from __future__ import annotations

import numpy as np


def linear_array(numel: int, pitch: float, offset=(0.0, 0.0, 0.0)):
    """Element positions ``3 x N`` and normals of a linear array.

    reference ``src/TransducerArray.m:95-99`` (``x = linspace(-w/2, w/2, numel)``) and ``:101-109``
    (normals ``[sin(theta); 0; cos(theta)]`` with ``theta = rot(1) = 0``).
    """
    w = (numel - 1) * pitch
    x = np.linspace(-w / 2, w / 2, numel)
    p = np.stack([x, np.zeros(numel), np.zeros(numel)]) + np.asarray(offset, float).reshape(3, 1)
    n = np.stack([np.zeros(numel), np.zeros(numel), np.ones(numel)])
    return p, n


def convex_array(numel: int, radius: float, angular_pitch_deg: float, offset=(0.0, 0.0, 0.0)):
    """Positions / normals of a curvilinear array; apex ("center") at ``offset - [0,0,radius]``.

    reference ``src/TransducerConvex.m:85-92`` (``R [sin t; 0; cos t] - [0;0;R]``) and ``:94-102``.
    """
    aw = (numel - 1) * angular_pitch_deg
    th = np.deg2rad(np.linspace(-aw / 2, aw / 2, numel))
    n = np.stack([np.sin(th), np.zeros(numel), np.cos(th)])
    p = radius * n - np.array([[0.0], [0.0], [radius]]) + np.asarray(offset, float).reshape(3, 1)
    return p, n


def scan_cartesian(x, z, y=(0.0,)):
    """Pixel positions ``3 x I1 x I2 x I3`` of a Cartesian scan in the reference's default order
    ``'ZXY'`` -- depth z is the FASTEST image axis (reference ``src/ScanCartesian.m:11,126-143``,
    ``src/Scan.m:194``)."""
    x, y, z = (np.asarray(v, float).reshape(-1) for v in (x, y, z))
    Z, X, Y = np.meshgrid(z, x, y, indexing="ij")
    return np.stack([X, Y, Z])


def scan_polar(r, a_deg, origin=(0.0, 0.0, 0.0), y=(0.0,)):
    """Pixel positions of a polar scan, order ``'RAY'``: ``z = r cos a``, ``x = r sin a`` about
    ``origin`` (reference ``src/ScanPolar.m:11,99-115``)."""
    r, a, y = (np.asarray(v, float).reshape(-1) for v in (r, a_deg, y))
    R, A, Y = np.meshgrid(r, np.deg2rad(a), y, indexing="ij")
    og = np.asarray(origin, float)
    return np.stack([R * np.sin(A) + og[0], Y + og[1], R * np.cos(A) + og[2]])


def sequence_args(seq_type: str, *, tx_pos=None, tx_normals=None, focus=None, tx_offset=(0.0, 0.0, 0.0)):
    """``(Pv, Nv, option_strings)`` exactly as ``UltrasoundSystem.DAS`` derives them from the
    sequence type (reference ``src/UltrasoundSystem.m:3340-3352``):

    * ``'FSA'``: ``Pv`` = transmit element positions, ``Nv`` = element normals, ``'diverging-waves'``
    * ``'PW'`` : ``Pv = [0;0;0]``, ``Nv = focus`` (unit normals), ``'plane-waves'``
    * ``'FC'|'VS'|'DV'``: ``Pv = focus``, ``Nv = normalize(focus - tx.offset)``; ``'DV'`` adds
      ``'diverging-waves'``
    """
    if seq_type == "FSA":
        return np.asarray(tx_pos, float), np.asarray(tx_normals, float), ["diverging-waves"]
    if seq_type == "PW":
        return np.zeros((3, 1)), np.asarray(focus, float), ["plane-waves"]
    if seq_type in ("FC", "VS", "DV"):
        f = np.asarray(focus, float)
        nf = f - np.asarray(tx_offset, float).reshape(3, 1)
        # NB the reference divides by norm(nf) of the whole 3 x M matrix (:3350); per-column
        # normalisation is the documented intent and identical for the sign test it feeds.
        nf = nf / np.linalg.norm(nf, axis=0, keepdims=True)
        return f, nf, (["diverging-waves"] if seq_type == "DV" else [])
    raise ValueError(f"unknown sequence type {seq_type!r}")


def point_target_data(scat_pos, scat_amp, Pr, Pv, Nv, *, VS=True, DV=False, c0=1540.0, fs, fc, T, t0=0.0,
                      bw=1.2, baseband=False, dtype=np.complex64):
    """Analytic echoes of point targets: ``x[t,n,m] = sum_k A_k g(t/fs + t0 - tau_k(n,m))`` with
    ``g(t) = exp(-(t fc bw)^2) exp(2j pi fc t)`` and ``tau_k = (dv + dr)/c0`` using the same time-of-flight
    model as the beamformer (reference ``src/bf.cu:104-114``).  A stand-in for ``greens``:
    synthetic-data generator for tests and benchmarks, NOT a simulator port."""
    P = np.asarray(scat_pos, float).reshape(3, -1)
    A = np.broadcast_to(np.asarray(scat_amp, float).reshape(-1), (P.shape[1],))
    Pr, Pv, Nv = (np.asarray(v, float).reshape(3, -1) for v in (Pr, Pv, Nv))
    N, M = Pr.shape[1], max(Pv.shape[1], Nv.shape[1])
    Pv = np.broadcast_to(Pv, (3, M)) if Pv.shape[1] == 1 else Pv
    Nv = np.broadcast_to(Nv, (3, M)) if Nv.shape[1] == 1 else Nv
    t0 = np.broadcast_to(np.asarray(t0, float).reshape(-1), (M,))
    t = np.arange(T)[:, None, None] / fs + t0[None, None, :]
    x = np.zeros((T, N, M), np.complex128)
    for k in range(P.shape[1]):
        rv = P[:, k:k + 1] - Pv
        if VS:
            r = np.linalg.norm(rv, axis=0)
            dv = r if DV else np.copysign(r, (rv * Nv).sum(0))
        else:
            dv = (rv * Nv).sum(0)
        dr = np.linalg.norm(P[:, k:k + 1] - Pr, axis=0)
        tau = (dv[None, :] + dr[:, None]) / c0             # N x M
        d = t - tau[None]
        g = np.exp(-(d * fc * bw) ** 2) * np.exp(2j * np.pi * fc * d)
        if baseband:                                       # downmixed at fc: x * exp(-2j pi fc t_abs)
            g = g * np.exp(-2j * np.pi * fc * t)
        x += A[k] * g
    return x.astype(dtype)
