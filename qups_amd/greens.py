"""Host side of the point-scatterer simulator (reference ``UltrasoundSystem.greens``, src/UltrasoundSystem.m:463-882; device
kernel src/greens.cu).  ``greens_kernel`` mirrors the kernel call (``k.feval`` at :718); ``greens`` mirrors the part of the
wrapper that builds the time axis and calls it for a full-synthetic-aperture acquisition (every transmit element fires alone;
``focusTx`` -- the reference's final step that combines these traces for focused / plane-wave sequences -- is a separate path and
not part of this entry).  The transmit-receive waveform (the reference convolves pulse and impulse responses, :603-606) is an
input here: samples, start time, sampling frequency.  There is no CPU fallback: the result comes from ``libqdas.so``."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _torch():
    import torch
    return torch


def greens_kernel(Ps, a, Pr, Pv, x, S, s0, t0, fs, fsr, cinv, R0, interp="cubic", prec="single", device=None):
    """``y (S x N x M complex)`` on the device.  ``Ps`` 3 x I, ``a`` (I,), ``Pr`` 3 x N [x En], ``Pv`` 3 x M [x Em], ``x`` (T,)."""
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError("qups_amd: no HIP device visible -- greens has no CPU fallback")
    if interp not in _lib.INTERP_FLAGS:
        raise ValueError(f"Interp option not recognized: {interp}")
    if prec not in ("single", "double"):
        raise ValueError("greens: prec must be 'single' or 'double'")
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    rt, ct = (np.float32, np.complex64) if prec == "single" else (np.float64, np.complex128)
    Ps = np.asarray(Ps, np.float64).reshape(3, -1)
    Pr = np.asarray(Pr, np.float64)
    Pv = np.asarray(Pv, np.float64)
    Pr = Pr.reshape(3, Pr.shape[1], -1)
    Pv = Pv.reshape(3, Pv.shape[1], -1)
    I, (N, En), (M, Em) = Ps.shape[1], Pr.shape[1:], Pv.shape[1:]
    col = lambda A, t: torch.from_numpy(np.ascontiguousarray(np.asarray(A).reshape(-1, order="F").astype(t))).to(dev)
    bufs = [col(Ps, rt), col(np.asarray(a).reshape(-1), ct), col(Pr, rt), col(Pv, rt), col(np.asarray(x).reshape(-1), ct)]
    T = int(np.asarray(x).size)
    y = torch.empty((M, N, int(S)), dtype=torch.complex64 if prec == "single" else torch.complex128, device=dev)      # (every sample is written: qdas_greens zero-fills the empty cases itself)
    d = _lib.GreensDesc()
    d.S, d.T, d.N, d.M, d.I = int(S), T, N, M, I
    d.En, d.Em, d.interp, d.dtype = En, Em, _lib.INTERP_FLAGS[interp], 1 if prec == "single" else 0
    d.s0, d.t0, d.fs, d.fsr, d.cinv, d.R0 = float(s0), float(t0), float(fs), float(fsr), float(cinv), float(R0)
    d.Ps, d.a, d.Pr, d.Pv, d.x = (C.c_void_p(b.data_ptr()) for b in bufs)
    d.device = dev.index if dev.index is not None else torch.cuda.current_device()
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().qdas_greens(C.byref(d), C.c_void_p(y.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return y.permute(2, 1, 0)                              # S x N x M view of the column-major buffer


def greens(rx_pos, tx_pos, scat_pos, scat_amp, c0, waveform, wv_t0, wv_fs, fs, R0=None, interp="cubic", prec="single",
           rx_bounds=None, tx_bounds=None, device=None):
    """Full-synthetic-aperture channel data of point scatterers: ``(data S x N x M, t0)`` with ``t0`` the time of sample 0.

    Time axis as the reference builds it (src/UltrasoundSystem.m:584-615): from the earliest / latest two-way path between the
    scatterers and the corners of the apertures' bounding boxes, padded by the waveform's duration."""
    from . import geometry  # noqa: F401  (geometry helpers live next to this module)
    rx_pos = np.asarray(rx_pos, np.float64).reshape(3, -1)
    tx_pos = np.asarray(tx_pos, np.float64).reshape(3, -1)
    Ps = np.asarray(scat_pos, np.float64).reshape(3, -1)
    wv = np.asarray(waveform).reshape(-1)
    dur = (wv.size - 1) / wv_fs
    bnd = lambda p, b: b if b is not None else [(p[k].min(), p[k].max()) for k in range(3)]
    corners = lambda b: np.array([[b[0][i & 1], b[1][(i >> 1) & 1], b[2][(i >> 2) & 1]] for i in range(8)]).T
    txb, rxb = corners(bnd(tx_pos, tx_bounds)), corners(bnd(rx_pos, rx_bounds))
    def dist_range(p):          # (min, max) distance between any scatterer and any corner: corner by corner, no I x 8 x 3 temporaries (100 000 scatterers: 2 ms, not 30)
        lo, hi = np.inf, 0.0
        for c in p.T:
            d2 = (Ps[0] - c[0]) ** 2
            d2 += (Ps[1] - c[1]) ** 2
            d2 += (Ps[2] - c[2]) ** 2
            lo, hi = min(lo, float(d2.min())), max(hi, float(d2.max()))
        return np.sqrt(lo), np.sqrt(hi)
    rng = lambda p: np.linalg.norm(p.max(1) - p.min(1))
    (tmin, tmax), (rmin, rmax) = dist_range(txb), dist_range(rxb)
    taumax = (tmax + rmax + rng(txb) + rng(rxb)) / c0
    taumin = (tmin + rmin - rng(txb) - rng(rxb)) / c0
    n0 = int(np.floor((taumin + wv_t0 - dur) * fs))
    ne = int(np.ceil((taumax + wv_t0 + dur) * fs))
    S = ne - n0 + 1
    if R0 is None:
        R0 = 0.0
    y = greens_kernel(Ps, scat_amp, rx_pos, tx_pos, wv, S, n0 / fs, wv_t0, fs, wv_fs / fs, 1.0 / c0, R0, interp, prec, device)
    return y, n0 / fs
