"""CPU oracle of the point-scatterer simulator ``greens`` -- TEST INFRASTRUCTURE ONLY, never imported by
``qups_amd``.  float64 numpy restatement of the reference kernel ``greens_temp`` (reference src/greens.cu:8-86) with the
time-axis / window marshalling of ``UltrasoundSystem.greens`` (src/UltrasoundSystem.m:584-615, 640-718), and the CPU
branch's R0 == 0 rule (src/UltrasoundSystem.m:797-803).  The interpolators are the DAS oracle's (``das_oracle.sample``).

Parity status: the reference pins ``greens`` only through integration tests (test/BFTest.m:124,306-316: a beamformed
image of greens() data peaks within 1.1 mm of the scatterer); the same criterion is applied in tests/test_greens.py.
"""
from __future__ import annotations

import numpy as np

from . import das_oracle as O


def greens_kernel(Ps, a, Pr, Pv, x, S, s0, t0, fs, fsr, cinv, R0, interp="cubic"):
    """y[s, n, m]  (S x N x M complex128).  Ps 3 x I, a (I,), Pr 3 x N x En, Pv 3 x M x Em, x (T,) complex."""
    Ps = np.asarray(Ps, np.float64).reshape(3, -1)
    a = np.asarray(a, np.complex128).reshape(-1)
    Pr = np.asarray(Pr, np.float64)
    Pv = np.asarray(Pv, np.float64)
    Pr = Pr.reshape(3, Pr.shape[1], -1)
    Pv = Pv.reshape(3, Pv.shape[1], -1)
    x = np.asarray(x, np.complex128).reshape(-1)
    N, En = Pr.shape[1:]
    M, Em = Pv.shape[1:]
    s = np.arange(S, dtype=np.float64)[:, None, None]
    y = np.zeros((S, N, M), np.complex128)
    for i in range(Ps.shape[1]):                                   # src/greens.cu:52
        for me in range(Em):                                       # :55
            for ne in range(En):                                   # :57
                r1 = np.linalg.norm(Ps[:, i, None] - Pr[:, :, ne], axis=0)[None, :, None]   # :61
                r2 = np.linalg.norm(Ps[:, i, None] - Pv[:, :, me], axis=0)[None, None, :]   # :62
                tau = s - (cinv * (r1 + r2) + t0 - s0) * fs                                # :65
                if R0:                                                                     # :68-74
                    g = np.maximum(r1, R0) * np.maximum(r2, R0)
                else:
                    g = 1.0
                y += a[i] * O.sample(x[:, None, None], fsr * tau, interp) / g               # :79
    return y / fsr                                                                         # :84 (R0^2 folded into g)


def time_axis(Ps, c0, tx_bounds, rx_bounds, wv_t0, wv_tend, wv_duration, fs):
    """output time axis of the wrapper (src/UltrasoundSystem.m:584-615): sample indices n0..ne at fs."""
    Ps = np.asarray(Ps, np.float64).reshape(3, -1)
    corners = lambda b: np.array([[b[0][i & 1], b[1][(i >> 1) & 1], b[2][(i >> 2) & 1]] for i in range(8)]).T   # :592-594
    txb, rxb = corners(tx_bounds), corners(rx_bounds)
    dist = lambda p: np.linalg.norm(Ps[:, :, None] - p[:, None, :], axis=0)
    rng = lambda p: np.linalg.norm(p.max(1) - p.min(1))
    taumax = (dist(txb).max() + dist(rxb).max() + rng(txb) + rng(rxb)) / c0                # :600
    taumin = (dist(txb).min() + dist(rxb).min() - rng(txb) - rng(rxb)) / c0                # :601
    tmin = taumin + wv_t0 - wv_duration                                                    # :608
    tmax = taumax + wv_tend                                                                # :609
    return int(np.floor(tmin * fs)), int(np.ceil(tmax * fs))                               # :611-612
