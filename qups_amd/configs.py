"""Input builders for the five BASELINE.json configurations (geometry per SURVEY.md section 8d).

These are synthetic inputs for ``bench.py`` and the full-size parity tests -- host-side numpy only.  Every
builder returns a dict with ``Pi, Pr, Pv, Nv`` (float64 arrays holding float32-exact values), the ``das_spec``
option strings, ``T, N, M, fs, c0, interp``, the image size and a human-readable label.
"""
from __future__ import annotations

import numpy as np

from . import geometry as G


def _f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def workload(name: str) -> dict:
    c0 = 1540.0
    apod = None
    rx_apod = None
    prec = "single"
    if name in ("c1", "c1f"):   # 64-el linear array (L7-4-like), 32 focused transmits, 256 x 256, linear interp ("c1f": foci INSIDE the image)
        fc, N, pitch, nx, nz, T, interp = 5.208e6, 64, 0.298e-3, 256, 256, 2048, "linear"
        lam = c0 / fc
        Pr, nrm = G.linear_array(N, pitch)
        zf = 30e-3 if name == "c1" else 12e-3
        xf = np.linspace(Pr[0, 8], Pr[0, -9], 32)                       # walking aperture, foci at z = 30 mm
        Pv, Nv, opt = G.sequence_args("FC", focus=np.stack([xf, 0 * xf, 0 * xf + zf]))
        x = (np.arange(nx) - (nx - 1) / 2) * lam / 4
        z = 2e-3 + np.arange(nz) * lam / 4
        Pi = G.scan_cartesian(x, z)
        t0 = -zf / c0 - 2e-6                                            # t = 0 when the wavefront passes the focus
        label = f"C1: 64-el linear array, 32 focused Tx (FC, foci at {zf * 1e3:g} mm), 256x256 ScanCartesian lambda/4, T=2048, linear"
    elif name == "c2":    # 128 el, 128 plane waves, 512^2, cubic
        fc, N, pitch, nx, nz, T, interp = 5e6, 128, 0.3e-3, 512, 512, 2048, "cubic"
        lam = c0 / fc
        Pr, nrm = G.linear_array(N, pitch)
        th = np.deg2rad(np.linspace(-25, 25, 128))
        Pv, Nv, opt = G.sequence_args("PW", focus=np.stack([np.sin(th), 0 * th, np.cos(th)]))
        x = (np.arange(nx) - (nx - 1) / 2) * lam / 4
        z = 2e-3 + np.arange(nz) * lam / 4
        Pi = G.scan_cartesian(x, z)
        t0 = -10e-3 / c0                                                 # steered waves reach the near corners before t = 0
        label = "C2: 128-el, 128 plane waves, 512x512 ScanCartesian lambda/4, T=2048, complex64, cubic"
    elif name.startswith("pw"):   # "pw9": the C2 setup with a handful of plane waves (ultrafast compounding: what a kHz-rate scanner beamforms)
        M = int(name[2:] or 9)
        fc, N, pitch, nx, nz, T, interp = 5e6, 128, 0.3e-3, 512, 512, 2048, "cubic"
        lam = c0 / fc
        Pr, nrm = G.linear_array(N, pitch)
        th = np.deg2rad(np.linspace(-12, 12, M))
        Pv, Nv, opt = G.sequence_args("PW", focus=np.stack([np.sin(th), 0 * th, np.cos(th)]))
        x = (np.arange(nx) - (nx - 1) / 2) * lam / 4
        z = 2e-3 + np.arange(nz) * lam / 4
        Pi = G.scan_cartesian(x, z)
        t0 = -10e-3 / c0
        label = f"PW{M}: 128-el, {M} plane waves, 512x512 ScanCartesian lambda/4, T=2048, complex64, cubic"
    elif name == "c3":    # 256 el FSA, 1024^2, lanczos3 (README headline)
        fc, N, pitch, nx, nz, T, interp = 5e6, 256, 0.2e-3, 1024, 1024, 2816, "lanczos3"
        lam = c0 / fc
        Pr, nrm = G.linear_array(N, pitch)
        Pv, Nv, opt = G.sequence_args("FSA", tx_pos=Pr, tx_normals=nrm)
        x = (np.arange(nx) - (nx - 1) / 2) * lam / 4
        z = 2e-3 + np.arange(nz) * lam / 4
        Pi = G.scan_cartesian(x, z)
        t0 = 0.0
        label = "C3: 256-el FSA (256 Tx x 256 Rx), 1024x1024 ScanCartesian lambda/4, T=2816, complex64, lanczos3"
    elif name == "c5":    # C5-2v convex, 96 diverging transmits, polar scan, fp16 data, per-pixel Rx apodization
        fc, N, R, ap, T, interp = 3.7e6, 128, 49.57e-3, 0.5872, 3072, "cubic"     # reference src/TransducerConvex.m:351-362
        nr, na = 512, 1024
        Pr, nrm = G.convex_array(N, R, ap)
        apex = np.array([0.0, 0.0, -R])
        th = np.deg2rad(np.linspace(-35, 35, 96))
        rv = R - 10e-3                                                  # virtual sources 10 mm behind the face
        Pv = apex[:, None] + rv * np.stack([np.sin(th), 0 * th, np.cos(th)])
        Pv, Nv, opt = G.sequence_args("DV", focus=Pv, tx_offset=apex)
        Pi = G.scan_polar(np.linspace(R, R + 150e-3, nr), np.linspace(-37, 37, na), origin=apex)
        # acceptance-angle receive apodization, I1 x I2 x 1 x N (reference src/UltrasoundSystem.m:5355-5373)
        d = Pi[:, :, :, 0, None] - Pr[:, None, None, :]
        cosang = (d * nrm[:, None, None, :]).sum(0) / np.maximum(np.linalg.norm(d, axis=0), 1e-12)
        apod = (cosang >= np.cos(np.deg2rad(30.0))).astype(np.float32)[:, :, None, :, None]
        rx_apod = ("acceptance", dict(theta=30.0))                      # the same rule, for generation inside the kernel
        nx, nz = na, nr
        t0 = -rv / c0 * 0                                               # DV: distance measured from the virtual source
        prec = "halfT"
        label = "C5: C5-2v convex (128 el), 96 diverging Tx, 512x1024 ScanPolar, T=3072, fp16 data, I1xI2x1xN Rx apod, cubic"
    elif name == "small":  # quick plumbing check
        fc, N, pitch, nx, nz, T, interp = 5e6, 32, 0.3e-3, 128, 256, 1024, "lanczos3"
        lam = c0 / fc
        Pr, nrm = G.linear_array(N, pitch)
        Pv, Nv, opt = G.sequence_args("FSA", tx_pos=Pr, tx_normals=nrm)
        x = (np.arange(nx) - (nx - 1) / 2) * lam / 4
        z = 2e-3 + np.arange(nz) * lam / 4
        Pi = G.scan_cartesian(x, z)
        t0 = 0.0
        label = "small: 32-el FSA, 256x128, T=1024, complex64, lanczos3"
    else:
        raise ValueError(f"unknown workload {name}")
    M = max(Nv.shape[1], Pv.shape[1])
    fs = float(np.float32(4 * fc))
    return dict(name=name, label=label, Pi=_f32(Pi), Pr=_f32(Pr), Pv=_f32(Pv), Nv=_f32(Nv), opt=list(opt), T=T, N=N, M=M, fs=fs,
                c0=c0, interp=interp, I1=nz, I2=nx, t0=float(np.float32(t0)), apod=apod, prec=prec,
                nrm=_f32(nrm), rx_apod=rx_apod)
