"""CPU restatement of the reference's ``convd`` (TEST INFRASTRUCTURE ONLY -- never imported by the product path).

Follows kern/convd.m: size checks and broadcasting (:63-84), output lags (:103-114: 'full' -(N-1)..M-1, 'same'
(0..M-1) - floor((N-1)/2), 'valid' 0..M-N), and the device kernel's sum (src/convd.cu:118-122)

    z[l] = sum_i x[i] * y[N-1 - (l0 - l + i)],   l0 = -lags[0]

which is the full convolution ``sum_i x[i] y[lf - i]`` read at ``lf = lags[l] + N - 1``.  Pinned (tests/test_oracle_pins.py) on the
reference's own documented example (kern/convd.m:33-35: ``convd([1 -2 3 -4 5], [5 -4 3 -2 1]) == xcorr([1 -2 3 -4 5])``), on the
MATLAB ``conv(u, v, shape)`` definition the reference's test compares against (test/KernTest.m:152-160), and on the default
``y = conj(flip(x))`` (kern/convd.m:55).  Computes in float64 / complex128."""
import numpy as np


def conv_lags(M, N, shape):
    if shape == "full":
        return np.arange(-(N - 1), M)
    if shape == "same":
        return np.arange(0, M) - (N - 1) // 2
    if shape == "valid":
        return np.arange(0, M - N + 1)
    raise ValueError(shape)


def convd(x, y=None, dim=None, shape="full"):
    """``dim`` is 1-based like the reference's.  Returns ``(z, lags)``."""
    x = np.asarray(x)
    if y is None:
        if dim is None:
            dim = next((k for k, v in enumerate(x.shape) if v != 1), 0) + 1
        xx = x.reshape(x.shape + (1,) * max(0, dim - x.ndim))
        y = np.conj(np.flip(xx, dim - 1))
    y = np.asarray(y)
    if dim is None:
        ds = [next((k for k, v in enumerate(a.shape) if v != 1), None) for a in (x, y)]
        ds = [d for d in ds if d is not None]
        dim = (min(ds) if ds else 0) + 1
    D = max(x.ndim, y.ndim, dim)
    x = x.reshape(x.shape + (1,) * (D - x.ndim))
    y = y.reshape(y.shape + (1,) * (D - y.ndim))
    d = dim - 1
    M, N = x.shape[d], y.shape[d]
    for k in range(D):
        if k != d and not (x.shape[k] == y.shape[k] or x.shape[k] == 1 or y.shape[k] == 1):
            raise ValueError("Incompatible sizes")
    full = [max(x.shape[k], y.shape[k]) for k in range(D)]
    fx, fy = list(full), list(full)
    fx[d], fy[d] = M, N
    cplx = np.iscomplexobj(x) or np.iscomplexobj(y)
    ct = np.complex128 if cplx else np.float64
    xb = np.moveaxis(np.broadcast_to(x, fx), d, -1).astype(ct).reshape(-1, M)
    yb = np.moveaxis(np.broadcast_to(y, fy), d, -1).astype(ct).reshape(-1, N)
    lags = conv_lags(M, N, shape)
    L = len(lags)
    z = np.zeros((xb.shape[0], L), ct)
    if L and M and N:
        for k in range(xb.shape[0]):
            f = np.convolve(xb[k], yb[k], "full")                       # f[lf] = sum_i x[i] y[lf - i]
            z[k] = f[lags + N - 1]
    osz = [full[k] for k in range(D) if k != d] + [L]
    z = np.moveaxis(z.reshape(osz), -1, d)
    lsz = [1] * D
    lsz[d] = L
    return z, lags.reshape(lsz)


def convd_direct(x, y, shape="full"):
    """1-D, the device kernel's loop literally (src/convd.cu:118-122) -- used to pin the vectorised path above."""
    x, y = np.asarray(x), np.asarray(y)
    M, N = len(x), len(y)
    lags = conv_lags(M, N, shape)
    l0 = -lags[0] if len(lags) else 0
    z = np.zeros(len(lags), np.result_type(x, y, np.float64))
    for l in range(len(lags)):
        i, j = 0, l0 - l
        while i < M or j < N:
            if 0 <= i < M and 0 <= j < N:
                z[l] += x[i] * y[N - 1 - j]
            i += 1
            j += 1
    return z


def sosfilt(x, sos, axis=0, gain=1.0):
    """IIR filtering as ``filter(D, x)`` applies an IIR ``digitalFilter`` (reference src/ChannelData.m:857-888 -> MATLAB ``filter``): the second-order sections
    ``[b0 b1 b2 a0 a1 a2]`` one after the other, each the direct-form II transposed recursion from rest
        y[t] = b0 x[t] + s1;  s1 = b1 x[t] - a1 y[t] + s2;  s2 = b2 x[t] - a2 y[t]        (coefficients / a0),
    ``gain`` applied once.  float64 (complex128), plain loops over time: TEST INFRASTRUCTURE (small cases; pinned against scipy.signal.sosfilt / lfilter in
    tests/test_convd.py)."""
    x = np.moveaxis(np.asarray(x), axis, 0)
    y = x.astype(np.complex128 if np.iscomplexobj(x) else np.float64) * gain
    for sec in np.asarray(sos, dtype=np.float64).reshape(-1, 6):
        b0, b1, b2, a0, a1, a2 = sec
        b0, b1, b2, a1, a2 = b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0
        s1 = np.zeros(y.shape[1:], dtype=y.dtype)
        s2 = np.zeros(y.shape[1:], dtype=y.dtype)
        out = np.empty_like(y)
        for t in range(y.shape[0]):
            xt = y[t]
            yt = b0 * xt + s1
            s1 = b1 * xt - a1 * yt + s2
            s2 = b2 * xt - a2 * yt
            out[t] = yt
        y = out
    return np.moveaxis(y, 0, axis)
