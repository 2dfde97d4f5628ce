"""CPU oracle for the QUPS delay-and-sum hot path (TEST INFRASTRUCTURE ONLY).

This file is a float64 numpy *restatement* of the reference algorithm.  It is
the checker for the HIP kernels in ``qups_amd/csrc``; nothing in the product
path (``qups_amd/``) may import it.  Only ``tests/``, ``__graft_entry__.smoke``
and the ``cpu_baseline`` leg of ``bench.py`` use it.

What it follows (paths relative to the reference checkout):

* geometry / time of flight ........ ``kern/das_spec.m:420-440,469`` (CPU branch)
                                      and ``src/bf.cu:104-114`` (device form)
* interpolators ..................... ``src/interpd.cu:68-150`` (nearest, linear,
                                      cubic = Keys a=-1/2 in Horner form,
                                      lanczos3 = window a=2, 4 taps)
* remodulation ...................... ``src/bf.cu:117`` (device semantics, default)
                                      or ``kern/das_spec.m:414-417`` (cpu variant)
* apodization / sound-speed bcast ... ``kern/das_spec.m:257-260,457-473``,
                                      ``src/bf.cu:87-90,113,120-123``
* accumulation modes ................ ``kern/das_spec.m:263-269,451-559``,
                                      ``src/bf.cu:129-140``
* transposed data (T x M x N) ....... ``src/bf.cu:100``, ``kern/das_spec.m:251,456``
* split-delay flavour ............... ``kern/wsinterpd2.m:276-295``,
                                      ``src/ChannelData.m:1431-1439``,
                                      ``src/interpd.cu:344-396``

Parity pinning status (see DESIGN.md "Oracle"):

* The reference is MATLAB + CUDA; neither can run in this environment and the
  reference ships NO stored golden vectors for this path.  ``nearest`` and
  ``linear`` are pinned against an independent restatement of MATLAB
  ``interp1(x, 1+tau, method, 0)`` (``numpy.interp`` / round-half-away) on the
  closed-form fixture of ``test/interpTest.m:33-43`` -- the same comparison that
  reference test performs (``test/interpTest.m:116-121,126,140``).
* ``cubic`` and ``lanczos3``: **parity unpinned by the reference** (its own
  test multiplies the cubic tolerance by 1e8..1e16, ``test/interpTest.m:127-133``,
  and only smoke-runs lanczos3, ``test/KernTest.m:189-193``).  The contract for
  both is the formula in ``src/interpd.cu:87-150``; this file restates it and is
  cross-checked against independent kernel-form (convolution) definitions.

Edge rule: a sample is in support iff ALL of its taps
are inside ``[0, T)`` (device rule, ``src/interpd.cu:72,84,93,138``) AND
``tau >= 0`` (MATLAB ``interp1(..., 0)`` rule); anything else is exactly 0.

Array conventions: MATLAB dimension ORDER is kept (``x`` is ``T x N x M x F``,
``Pi`` is ``3 x I1 x I2 x I3`` ...) but these are ordinary numpy arrays; memory
layout does not matter here.
"""
from __future__ import annotations

import numpy as np

# bits 0-2 of the kernel flag (kern/das_spec.m:198-203; 4 == linear on the device,
# src/interpd.cu:163); 5 is this build's opt-in extension (see interp_weights)
INTERP_FLAGS = {"nearest": 0, "linear": 1, "cubic": 2, "lanczos3": 3, "cubic_dev": 5}


# --------------------------------------------------------------------------
# interpolators (src/interpd.cu:68-150)
# --------------------------------------------------------------------------
def lanczos_helper(v: np.ndarray, a: int = 2) -> np.ndarray:
    """``L(v) = 2 sin(pi v) sin(pi v / a) / (pi^2 v^2)``, ``L(0) = 1``.

    Follows ``src/interpd.cu:116-127`` (note the window is a=2 despite the name
    "lanczos3", ``src/interpd.cu:134``).
    """
    v = np.asarray(v, dtype=np.float64)
    out = np.ones_like(v)
    nz = v != 0.0
    vn = v[nz]
    out[nz] = 2.0 * np.sin(np.pi * vn) * np.sin(np.pi * vn / a) / (np.pi**2 * vn * vn)
    return out


def interp_weights(u: np.ndarray, interp: str):
    """Tap weights for fractional offset ``u`` in [0,1).

    Returns ``(first_tap_offset, weights[K, ...])`` where taps are
    ``ti + first_tap_offset + k`` for ``k = 0..K-1`` and ``ti = floor(tau)``
    (for ``nearest`` the single tap is ``round(tau)`` and is handled by the
    caller).
    """
    u = np.asarray(u, dtype=np.float64)
    if interp == "linear":  # src/interpd.cu:77-85 ; lerp = a + t (b - a)
        return 0, np.stack([1.0 - u, u])
    if interp == "cubic":
        # Catmull-Rom == Keys cubic convolution a=-1/2 == MATLAB interp1 'cubic'
        # on a uniform grid (the CPU path, kern/das_spec.m:477) == the weights
        # the device source documents in its comment (src/interpd.cu:108-111):
        #   1/2 { -u^3+2u^2-u, 3u^3-5u^2+2, -3u^3+4u^2+u, u^3-u^2 }
        a0 = u * (-1.0 + u * (2.0 - u))
        a1 = 2.0 + u * u * (3.0 * u - 5.0)
        a2 = u * (1.0 + u * (4.0 - 3.0 * u))
        a3 = u * u * (u - 1.0)
        return -1, 0.5 * np.stack([a0, a1, a2, a3])
    if interp == "cubic_dev":
        # The Horner lines the device code actually EXECUTES (src/interpd.cu:103-106,
        # same in src/interpolators.cl:108-111) expand to a DIFFERENT cubic:
        #   1/2 { 2u^3-u^2-u, -5u^3+3u^2+2, 4u^3-3u^2+u, -u^3+u^2 }
        # (u^3 and u^2 coefficients swapped w.r.t. the comment).  It still
        # interpolates and reproduces linears, but is not Catmull-Rom -- which is
        # why test/interpTest.m:127-133 needs a 1e8 x tolerance for 'cubic'.
        # Kept as an opt-in, bug-compatible variant of the device path.
        a0 = 0.0 + u * (-1.0 + u * (+2.0 * u - 1.0))
        a1 = 2.0 + u * (+0.0 + u * (-5.0 * u + 3.0))
        a2 = 0.0 + u * (+1.0 + u * (+4.0 * u - 3.0))
        a3 = 0.0 + u * (+0.0 + u * (-1.0 * u + 1.0))
        return -1, 0.5 * np.stack([a0, a1, a2, a3])
    if interp == "lanczos3":  # src/interpd.cu:141-149
        return -1, np.stack(
            [lanczos_helper(u + 1), lanczos_helper(u), lanczos_helper(u - 1), lanczos_helper(u - 2)]
        )
    raise ValueError(f"Unrecognized interpolation of type {interp}")


def sample(x: np.ndarray, s: np.ndarray, interp: str) -> np.ndarray:
    """Sample trace(s) ``x`` (time on axis 0) at fractional 0-based indices ``s``.

    ``x`` has shape ``(T, *B)`` and ``s`` has shape ``(P, *B)`` (or broadcastable
    to it in the trailing axes): every pixel ``p`` samples every trace.  Returns
    ``(P, *B)`` complex128/float64.  Out of support -> exactly 0.
    """
    x = np.asarray(x)
    T = x.shape[0]
    s = np.asarray(s, dtype=np.float64)
    bshape = np.broadcast_shapes(s.shape[1:], x.shape[1:])
    s = np.broadcast_to(s, (s.shape[0],) + bshape)
    xb = np.broadcast_to(x, (T,) + bshape)
    cplx = np.iscomplexobj(x)
    out = np.zeros(s.shape, dtype=np.complex128 if cplx else np.float64)
    fin = np.isfinite(s)
    s0 = np.where(fin, s, -1.0)
    bidx = np.indices(s.shape)[1:]  # trailing-axes indices for fancy gather

    def gather(ti):
        return xb[(ti,) + tuple(bidx)]

    if interp == "nearest":  # src/interpd.cu:70-72 (roundf = half away from zero)
        ti = np.floor(s0 + 0.5).astype(np.int64)  # s0 >= 0 wherever valid
        valid = fin & (s0 >= 0.0) & (ti < T)
        tic = np.clip(ti, 0, T - 1)
        out = np.where(valid, gather(tic), 0.0)
        return out
    ti = np.floor(s0).astype(np.int64)
    u = s0 - ti
    off, w = interp_weights(u, interp)
    K = w.shape[0]
    first = ti + off
    valid = fin & (s0 >= 0.0) & (first >= 0) & (first + K - 1 < T)
    acc = np.zeros(s.shape, dtype=out.dtype)
    for k in range(K):
        tk = np.clip(first + k, 0, T - 1)
        acc = acc + w[k] * gather(tk)
    return np.where(valid, acc, 0.0)


# --------------------------------------------------------------------------
# geometry (kern/das_spec.m:420-440 ; src/bf.cu:104-110)
# --------------------------------------------------------------------------
def _as3(P):
    """Coordinates to 3 x ... (kern/das_spec.m:649-669: 1D->x, 2D->(x,z), 4D->xyz/w)."""
    P = np.asarray(P, dtype=np.float64)
    d = P.shape[0]
    if d == 3:
        return P
    z = np.zeros((1,) + P.shape[1:])
    if d == 1:
        return np.concatenate([P, z, z], 0)
    if d == 2:
        return np.concatenate([P[:1], z, P[1:2]], 0)
    if d == 4:
        return P[:3] / P[3:4]
    raise ValueError("Improper coordinate dimension.")


def tx_rx_distances(Pi, Pr, Pv, Nv, VS=True, DV=False):
    """Return ``dv`` (I1,I2,I3,1,M) and ``dr`` (I1,I2,I3,N,1).

    ``kern/das_spec.m:427-436``: virtual source -> ``|Pi-Pv| * sign((Pi-Pv).Nv)``
    (sign := +1 for diverging waves), plane wave -> ``(Pi-Pv).Nv``.
    The device form uses ``copysign`` (``src/bf.cu:106-108``), which differs from
    ``sign`` only where the dot product is exactly 0 (sign -> 0 on the CPU
    branch, +|r| on the device); the oracle follows the DEVICE form.
    """
    Pi = _as3(Pi)
    while Pi.ndim < 4:
        Pi = Pi[..., None]
    Pr = _as3(Pr).reshape(3, -1)
    Pv = _as3(Pv).reshape(3, -1)
    Nv = _as3(Nv).reshape(3, -1)
    M = max(Pv.shape[1], Nv.shape[1])
    if Pv.shape[1] == 1:
        Pv = np.repeat(Pv, M, 1)
    if Nv.shape[1] == 1:
        Nv = np.repeat(Nv, M, 1)
    rv = Pi[..., None] - Pv[:, None, None, None, :]  # 3 x I1 x I2 x I3 x M
    if VS:
        r = np.sqrt((rv * rv).sum(0))
        if DV:
            dv = r
        else:
            dv = np.copysign(r, (rv * Nv[:, None, None, None, :]).sum(0))
    else:
        dv = (rv * Nv[:, None, None, None, :]).sum(0)
    rr = Pi[..., None] - Pr[:, None, None, None, :]
    dr = np.sqrt((rr * rr).sum(0))  # I1 x I2 x I3 x N
    return dv[:, :, :, None, :], dr[:, :, :, :, None]


def _bcast5(a, Isz, N, M, what):
    a = np.asarray(a)
    while a.ndim < 5:
        a = a[..., None]
    full = tuple(Isz) + (N, M)
    if a.ndim > 5 or any(s not in (1, f) for s, f in zip(a.shape, full)):
        raise ValueError(f"{what} data size inconsistent with pixel/receiver/transmit data size")
    return a


# --------------------------------------------------------------------------
# the beamformer (kern/das_spec.m:391-560 ; src/bf.cu:49-142)
# --------------------------------------------------------------------------
def das_spec(fun, Pi, Pr, Pv, Nv, x, t0, fs, c=1540.0, *, VS=True, DV=False,
             interp="linear", apod=(), fmod=0.0, tpose=False,
             fmod_mode="device"):
    """Float64 restatement of ``das_spec`` (see module docstring for the map).

    Returns ``y`` shaped ``I1 x I2 x I3 x [1|N] x [1|M] x F...`` exactly like
    ``kern/das_spec.m:381`` (``delays``: ``I1 x I2 x I3 x N x M`` real).
    """
    if fun not in ("DAS", "SYN", "MUL", "BF", "delays"):
        raise ValueError("Invalid beamformer.")
    Pi = _as3(Pi)
    while Pi.ndim < 4:
        Pi = Pi[..., None]
    Isz = Pi.shape[1:4]
    dv, dr = tx_rx_distances(Pi, Pr, Pv, Nv, VS, DV)
    N, M = dr.shape[3], dv.shape[4]
    cinv = _bcast5(1.0 / np.asarray(c, dtype=np.float64), Isz, N, M, "Sound speed")
    if fun == "delays":
        return cinv * (dv + dr)

    x = np.asarray(x)
    if tpose:  # data is T x M x N x F...  (src/bf.cu:100)
        x = np.swapaxes(x, 1, 2)
    if x.ndim < 3:
        x = x.reshape(x.shape + (1,) * (3 - x.ndim))
    T = x.shape[0]
    if x.shape[1] != N:
        raise ValueError("Inconsistent receiver data size.")
    if x.shape[2] != M:
        raise ValueError("Inconsistent transmitter data size.")
    fsz = x.shape[3:]
    xf = x.reshape(T, N, M, -1).astype(np.complex128)
    F = xf.shape[3]
    t0 = np.broadcast_to(np.asarray(t0, dtype=np.float64).reshape(-1), (M,)) if np.size(t0) in (1, M) \
        else (_ for _ in ()).throw(ValueError("t0 must be a scalar or have one value per transmit"))
    apods = [_bcast5(a, Isz, N, M, "Apodization") for a in apod]

    if fmod and fmod_mode == "cpu":  # kern/das_spec.m:414-417 (pre-interp, absolute t)
        if np.ptp(t0) != 0:
            raise ValueError("cpu fmod variant is defined for scalar t0")
        t = t0[0] + np.arange(T) / fs
        xf = xf * np.exp(2j * np.pi * fmod * t)[:, None, None, None]

    keep_rx = fun in ("SYN", "BF")
    keep_tx = fun in ("MUL", "BF")
    y = np.zeros(tuple(Isz) + (N if keep_rx else 1, M if keep_tx else 1, F), np.complex128)
    I = int(np.prod(Isz))
    for m in range(M):
        dvm = dv[..., 0, m].reshape(I)
        for n in range(N):
            cin = cinv[..., min(n, cinv.shape[3] - 1), min(m, cinv.shape[4] - 1)]
            cin = np.broadcast_to(cin, Isz).reshape(I)
            tau = cin * (dvm + dr[..., n, 0].reshape(I)) - t0[m]  # kern/das_spec.m:469
            val = sample(xf[:, n, m, :], (tau * fs)[:, None], interp)  # I x F
            if fmod and fmod_mode == "device":  # src/bf.cu:117
                val = val * np.exp(2j * np.pi * fmod * tau)[:, None]
            for a in apods:
                an = a[..., min(n, a.shape[3] - 1), min(m, a.shape[4] - 1)]
                val = val * np.broadcast_to(an, Isz).reshape(I, 1)
            val = val.reshape(tuple(Isz) + (F,))
            y[:, :, :, n if keep_rx else 0, m if keep_tx else 0, :] += val
    if tpose and keep_rx and keep_tx:   # 'BF' output keeps the DATA's aperture order (src/bf.cu:100,135)
        y = np.swapaxes(y, 3, 4)
    return y.reshape(tuple(Isz) + y.shape[3:5] + tuple(fsz))


# --------------------------------------------------------------------------
# split-delay flavour: bfDASLUT -> sample2sep -> wsinterpd2
# (src/UltrasoundSystem.m:4641-4660 ; src/ChannelData.m:1431-1445 ;
#  kern/wsinterpd2.m:286-290 ; src/interpd.cu:389-393)
# --------------------------------------------------------------------------
def das_lut(x, tau_rx, tau_tx, t0, fs, *, interp="cubic", apod=(), fmod=0.0,
            keep_rx=False, keep_tx=False):
    """``y[i,(n),(m),f] = sum w * exp(2j pi fmod/fs * s) * sample(x[:,n,m,f], s)``
    with ``s = (tau_rx[i,n] + tau_tx[i,m] - t0[m]) * fs``.

    ``x``: T x N x M x F...; ``tau_rx``: I.. x N; ``tau_tx``: I.. x M (leading
    pixel dims identical).  Non-finite delays are skipped (``src/interpd.cu:390``).
    The phasor uses the SAMPLE-index delay (``omega = 2i pi fmod / fs`` times
    ``ntau``, ``src/ChannelData.m:1439``), i.e. time relative to ``t0``.
    """
    x = np.asarray(x)
    T, N, M = x.shape[:3]
    fsz = x.shape[3:]
    xf = x.reshape(T, N, M, -1).astype(np.complex128)
    F = xf.shape[3]
    tau_rx = np.asarray(tau_rx, dtype=np.float64)
    tau_tx = np.asarray(tau_tx, dtype=np.float64)
    Isz = tau_rx.shape[:-1]
    I = int(np.prod(Isz))
    trx = tau_rx.reshape(I, N)
    ttx = tau_tx.reshape(I, M)
    t0 = np.broadcast_to(np.asarray(t0, dtype=np.float64).reshape(-1), (M,))
    Isz3 = tuple(Isz) + (1,) * (3 - len(Isz))
    apods = [_bcast5(a, Isz3, N, M, "Apodization") for a in apod]
    y = np.zeros((I, N if keep_rx else 1, M if keep_tx else 1, F), np.complex128)
    for m in range(M):
        for n in range(N):
            s = (trx[:, n] + ttx[:, m] - t0[m]) * fs
            val = sample(xf[:, n, m, :], s[:, None], interp)
            if fmod:
                ph = np.exp(2j * np.pi * fmod / fs * np.where(np.isfinite(s), s, 0.0))
                val = val * ph[:, None]
            for a in apods:
                an = a[..., min(n, a.shape[3] - 1), min(m, a.shape[4] - 1)]
                val = val * np.broadcast_to(an, Isz3).reshape(I, 1)
            y[:, n if keep_rx else 0, m if keep_tx else 0, :] += val
    return y.reshape(tuple(Isz) + y.shape[1:3] + tuple(fsz))


# --------------------------------------------------------------------------
# independent cross-checks used to pin the oracle (tests/test_oracle_pins.py)
# --------------------------------------------------------------------------
def interp1_matlab(x, xq, method):
    """Independent restatement of MATLAB ``interp1(x, xq, method, 0)`` for
    ``method in {'nearest','linear'}`` on the default grid 1..T (1-based ``xq``).

    Used the way ``test/interpTest.m:118`` uses ``interp1`` -- as the expected
    value for the kernels.  NOT built on :func:`sample`.
    """
    x = np.asarray(x)
    T = x.shape[0]
    xq = np.asarray(xq, dtype=np.float64)
    grid = np.arange(1, T + 1, dtype=np.float64)
    inside = (xq >= 1) & (xq <= T)
    if method == "linear":
        re = np.interp(xq, grid, x.real, left=0.0, right=0.0)
        if np.iscomplexobj(x):
            return re + 1j * np.interp(xq, grid, x.imag, left=0.0, right=0.0)
        return re
    if method == "nearest":  # MATLAB rounds half away from zero (up, for xq>0)
        k = np.floor(np.where(inside, xq, 1.0) + 0.5).astype(np.int64) - 1
        k = np.clip(k, 0, T - 1)
        return np.where(inside, x[k], 0.0)
    raise ValueError(method)


def keys_kernel(v, a=-0.5):
    """Keys cubic-convolution kernel (kernel form, independent of the Horner
    weights in :func:`interp_weights`)."""
    v = np.abs(np.asarray(v, dtype=np.float64))
    return np.where(v <= 1, (a + 2) * v**3 - (a + 3) * v**2 + 1,
                    np.where(v < 2, a * v**3 - 5 * a * v**2 + 8 * a * v - 4 * a, 0.0))


def lanczos_kernel(v, a=2):
    """Lanczos kernel ``sinc(v) sinc(v/a)`` on |v| < a (kernel form)."""
    v = np.asarray(v, dtype=np.float64)
    return np.where(np.abs(v) < a, np.sinc(v) * np.sinc(v / a), 0.0)


def convolve_sample(x, s, kernel, support):
    """Direct convolution-form interpolation ``sum_k x[k] K(s - k)`` with the
    all-taps-in-bounds rule; used only to cross-check :func:`sample`."""
    x = np.asarray(x)
    T = x.shape[0]
    s = np.asarray(s, dtype=np.float64)
    out = np.zeros(s.shape, dtype=np.complex128)
    for j, sj in np.ndenumerate(s):
        ti = int(np.floor(sj))
        ks = range(ti - support + 1, ti + support + 1)
        if sj < 0 or ks[0] < 0 or ks[-1] >= T:
            continue
        out[j] = sum(x[k] * kernel(sj - k) for k in ks)
    return out


# --------------------------------------------------------------------------
# Receive-apodization generators -- restated line by line from the reference, in the
# reference's own formulation (degrees, atan2d / sind / cosd), float64.  Pi: 3 x I1 x I2 x I3, Pr: 3 x N,
# nrm: 3 x N element normals, ang: N element azimuth angles in degrees.  Output I1 x I2 x I3 x N x 1.
# --------------------------------------------------------------------------
def _cosd(d):
    d = np.asarray(d, dtype=np.float64)
    r = np.cos(np.deg2rad(d))
    m = np.mod(d, 360.0)
    for a, v in ((0.0, 1.0), (60.0, 0.5), (90.0, 0.0), (120.0, -0.5), (180.0, -1.0), (240.0, -0.5), (270.0, 0.0), (300.0, 0.5)):
        r = np.where(m == a, v, r)                                 # MATLAB cosd is exact at these angles
    return r


def _sind(d):
    return _cosd(np.asarray(d, dtype=np.float64) - 90.0)


def ap_acceptance_angle(Pi, Pr, nrm, theta=45.0):
    """reference src/UltrasoundSystem.m:5355-5373"""
    Pi = np.asarray(Pi, np.float64)
    Pi = Pi.reshape(3, *Pi.shape[1:], *([1] * (4 - Pi.ndim)))
    Pn = np.asarray(Pr, np.float64).reshape(3, 1, 1, 1, -1)        # :5356
    n = np.asarray(nrm, np.float64).reshape(3, 1, 1, 1, -1)        # :5357-5358
    r = Pi[..., None] - Pn                                         # :5364
    with np.errstate(invalid="ignore", divide="ignore"):
        r = r / np.sqrt((r * r).sum(0, keepdims=True))             # :5369
        c = (n * r).sum(0)                                         # :5370
        return (c >= float(_cosd(theta))).astype(np.float64)[..., None]   # :5373


def ap_cosine_angle(Pi, Pr, nrm, theta=45.0):
    """reference src/UltrasoundSystem.m:5414-5428"""
    Pi = np.asarray(Pi, np.float64)
    Pi = Pi.reshape(3, *Pi.shape[1:], *([1] * (4 - Pi.ndim)))
    pn = np.asarray(Pr, np.float64).reshape(3, 1, 1, 1, -1)
    nn = np.asarray(nrm, np.float64).reshape(3, 1, 1, 1, -1)
    r = Pi[..., None] - pn                                         # :5421
    with np.errstate(invalid="ignore", divide="ignore"):
        r = r / np.sqrt((r * r).sum(0, keepdims=True))             # :5422
    r = (nn * r).sum(0)                                            # :5423
    r = np.fmax(-1.0, np.fmin(1.0, r))                             # :5424  (MATLAB max/min ignore NaN: NaN -> 1)
    return _cosd(np.minimum(90.0, (90.0 / theta) * np.rad2deg(np.arccos(r))))[..., None]   # :5425


def ap_aperture_growth(Pi, Pr, ang=None, f=1.5, Dmax=np.inf):
    """reference src/UltrasoundSystem.m:5227-5262 (generic-scan branch :5236-5239)"""
    Pi = np.asarray(Pi, np.float64)
    Pi = Pi.reshape(3, *Pi.shape[1:], *([1] * (4 - Pi.ndim)))
    Pr = np.asarray(Pr, np.float64)
    Xn, Zn = Pr[0].reshape(1, 1, 1, -1), Pr[2].reshape(1, 1, 1, -1)            # :5228-5229
    Xi, Zi = Pi[0][..., None], Pi[2][..., None]                                # :5237-5238
    if ang is not None and np.any(np.asarray(ang) != 0):                       # :5243
        ae = np.asarray(ang, np.float64).reshape(1, 1, 1, -1)                  # :5244
        rp = np.hypot(Xi - Xn, Zi - Zn)                                        # :5245
        ap = np.rad2deg(np.arctan2(Xi - Xn, Zi - Zn))                          # :5246
        d = rp * _sind(ap - ae)                                                # :5247
        z = np.abs(rp * _cosd(ap - ae))                                        # :5248
    else:
        d = Xn - Xi                                                            # :5251
        z = Zi - 0.0 + 0.0 * d                                                 # :5252
    apod = (z > f * np.abs(2 * d)).astype(np.float64)                          # :5255
    apod = apod * (np.abs(2 * d) < Dmax)                                       # :5256
    return apod[..., None]


# --------------------------------------------------------------------------
# general single-delay flavour: kern/wsinterpd.m (CPU branch :240-274), kern/interpd.m
# --------------------------------------------------------------------------
def wsinterpd(x, t, dim=1, w=1, sdim=None, interp="linear", extrapval=np.nan, omega=0):
    """``y = wsinterpd(x, t, dim, w, sdim, interp, extrapval, omega)`` restated with numpy broadcasting.

    After swapping ``dim`` with dimension 1 (``kern/wsinterpd.m:54-60``): ``x`` is ``T x ...``, ``t`` is ``I x ...``; every data
    dimension matches or is singleton in one of them (``:63-69``).  ``y = sum_sdim( exp(omega*t) .* w .* interp1(x, 1+t, interp,
    extrapval), 'omitnan')`` (``:262``) with the interpolators of ``src/interpd.cu:68-150`` (edge rule of this repository:
    a sample is in the record iff all taps are in ``[0, T)`` and ``t >= 0``).  Infinite ``t`` contribute nothing
    (``src/interpd.cu:333``)."""
    x, t = np.asarray(x), np.asarray(t, dtype=np.float64)
    wa = np.asarray(w)
    nd = max(x.ndim, t.ndim, wa.ndim, dim)
    pad = lambda a: a.reshape(a.shape + (1,) * (nd - a.ndim))
    sw = lambda a: np.swapaxes(a, 0, dim - 1)
    x, t, wa = sw(pad(x)), sw(pad(t)), sw(pad(wa))
    sd = [] if sdim is None else [int(v) for v in np.atleast_1d(sdim)]
    sd = [dim if v == 1 else (1 if v == dim else v) for v in sd]
    sd = [v for v in sd if v <= nd and not (x.shape[v - 1] == 1 and t.shape[v - 1] == 1)]      # kern/wsinterpd.m:73
    val = sample(x, t, interp)                                   # I x broadcast(data dims): out of record -> 0
    # in-record mask (for extrapval): recompute support like sample()
    T = x.shape[0]
    tb = np.broadcast_to(t, val.shape)
    fin = np.isfinite(tb)
    s0 = np.where(fin, tb, -1.0)
    if interp == "nearest":
        inrec = fin & (s0 >= 0) & (np.floor(s0 + 0.5) < T)
    else:
        off, wk = interp_weights(s0 - np.floor(s0), interp)
        first = np.floor(s0).astype(np.int64) + off
        inrec = fin & (s0 >= 0) & (first >= 0) & (first + wk.shape[0] - 1 < T)
    val = np.where(inrec, val, extrapval).astype(np.complex128)
    val = np.where(np.isinf(tb), 0.0, val)                      # infinite delays are skipped
    if omega != 0:
        val = val * np.exp(omega * np.where(fin, tb, 0.0))
    val = val * wa
    if sd:
        val = np.where(np.isnan(val), 0.0, val)                  # 'omitnan'
        val = val.sum(axis=tuple(v - 1 for v in sd), keepdims=True)
    return sw(val)


def interpd(x, t, dim=1, interp="linear", extrapval=np.nan):
    """``kern/interpd.m``: ``wsinterpd`` without weights / sums"""
    return wsinterpd(x, t, dim, 1, None, interp, extrapval, 0)


# --------------------------------------------------------------------------
# Sequence.delays / apodization (src/Sequence.m:888-1006), ChannelData.zeropad / rectifyt0 (src/ChannelData.m:1153-1228),
# UltrasoundSystem.focusTx (src/UltrasoundSystem.m:3374-3503)
# --------------------------------------------------------------------------
def sequence_delays(seq_type, tx_pos, focus=None, c0=1540.0):
    """``tau = seq.delays(tx)`` (N x S), ``src/Sequence.m:888-931``"""
    p = np.asarray(tx_pos, float)                                # 3 x N
    N = p.shape[1]
    if seq_type == "FSA":
        return np.zeros((N, N))
    f = np.asarray(focus, float)                                 # 3 x S
    if seq_type == "PW":
        return -(f[:, None, :] * p[:, :, None]).sum(0) / c0      # :914
    v = f[:, None, :] - p[:, :, None]                            # element -> focus, 3 x N x S
    tau = np.sqrt((v ** 2).sum(0)) / c0
    if seq_type == "FC":
        s = 1.0
    elif seq_type == "DV":
        s = -1.0
    elif seq_type == "VS":                                       # :908: behind the transducer -> negative
        s = np.where(np.all(f[2][None, :] > p[2][:, None], axis=0), 1.0, -1.0)[None, :]
    else:
        raise ValueError(seq_type)
    return tau * s


def sequence_apodization(seq_type, N, S):
    """``src/Sequence.m:953-975``"""
    return np.eye(N) if seq_type == "FSA" else np.ones((N, S))


def zeropad(x, t0, fs, B=0, A=0):
    """``src/ChannelData.m:1153-1183``: B zeros in front (t0 moves back), A behind; time is axis 0"""
    x = np.asarray(x)
    z = lambda n: np.zeros((n,) + x.shape[1:], x.dtype)
    return np.concatenate([z(B), x, z(A)], 0), np.asarray(t0, float) - B / fs


def rectifyt0(x, t0, fs, interp="cubic", index_dtype=np.float64):
    """``src/ChannelData.m:1205-1228`` for data ``T x N x M`` and ``t0`` of shape ``1 x 1 x M``: one start time for all transmits.
    (Mirrors the reference including its length: the time axis is extended by npad twice, ``:1220-1222``.)  ``index_dtype``:
    precision the sample indices are rounded to before sampling -- ``float32`` for single-precision data, as ``wsinterpd`` casts
    them (``kern/wsinterpd.m:128-130``); at whole-sample offsets the record edge depends on it."""
    t0 = np.asarray(t0, float)
    if t0.size == 1:
        return np.asarray(x), float(t0.reshape(-1)[0])
    t0_ = float(t0.min())
    npad = int(np.ceil((t0 - t0_).max() * fs))
    xp, _ = zeropad(x, t0, fs, 0, npad)
    T2 = xp.shape[0]
    tau = t0_ + np.arange(T2 + npad).reshape(-1, 1, 1) / fs
    ntau = ((tau - t0.reshape(1, 1, -1)) * fs).astype(index_dtype)      # the device kernels take the indices in the data precision
    y = wsinterpd(xp, ntau, 1, 1, None, interp, 0.0, 0)
    return y, t0_


def focus_tx(x, t0, fs, tau_seq, apd, interp="cubic", buffer=0):
    """``chd = focusTx(us, chd, seq)``: synthesise a sequence's transmits from full-synthetic-aperture data ``x`` (``T x N x M``).

    ``tau_seq = seq.delays(tx)`` (M x M'), ``apd = seq.apodization(tx)`` ([1|M] x [1|M']).  Reference
    ``src/UltrasoundSystem.m:3457-3502``: ``tau = -delays``; the time axis is shifted / extended to hold every delayed trace
    (``:3463-3470``); ``z[t', n, m'] = sum_m apd[m, m'] * x(time[t'] - tau[m, m'], n, m)`` via ``sample2sep`` (``:3498``)."""
    x = np.asarray(x)
    T, N, M = x.shape
    tau = -np.asarray(tau_seq, float)
    apd = np.broadcast_to(np.asarray(apd, float), tau.shape)
    i = apd != 0
    nmin = int(np.floor(np.nanmin(tau[i]) * fs))
    nmax = int(np.ceil(np.nanmax(tau[i]) * fs))
    t0n = float(np.asarray(t0).reshape(-1)[0]) + nmin / fs
    tau = tau - nmin / fs
    xp, _ = zeropad(x, t0n, fs, 0, (nmax - nmin) + int(buffer))
    T2 = xp.shape[0]
    Mp = tau.shape[1]
    z = np.zeros((T2, N, Mp), np.complex128)
    tt = np.arange(T2, dtype=np.float64)
    for mp in range(Mp):
        for m in range(M):
            if apd[m, mp] == 0:
                continue
            s = tt - tau[m, mp] * fs                              # (time - tau - t0) * fs
            z[:, :, mp] += apd[m, mp] * sample(xp[:, :, m], s[:, None], interp)
    return z, t0n
