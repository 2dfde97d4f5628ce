"""``convd`` -- batched convolution along one dimension on the device (reference kern/convd.m, kernels src/convd.cu:95-146).

Mirrors the reference's call ``[C, lags] = convd(x, y, dim, shape)``: ``x`` and ``y`` are N-D arrays with compatible sizes
(singleton dimensions broadcast) that are convolved along ``dim`` (1-based, as in the reference; default: the first
non-singleton dimension); ``shape`` is ``'full'`` (default), ``'same'`` or ``'valid'``; ``convd(x)`` is the auto-correlation
(``y = conj(flip(x))``, kern/convd.m:55).  The work is done by ``qdas_convd`` in ``libqdas.so`` (qups_amd/csrc/conv.hip); there is
no CPU fallback.  Torch tensors are row-major, so the dimensions BEHIND ``dim`` are the fast "columns" of the C ABI and the
ones in front of it the "slices": no transposition or replication of the data takes place."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

_SHAPES = {"full": _lib.QDAS_CONV_FULL, "same": _lib.QDAS_CONV_SAME, "valid": _lib.QDAS_CONV_VALID, "causal": _lib.QDAS_CONV_CAUSAL}


def conv_lags(M: int, N: int, shape: str):
    """lags of the outputs (reference kern/convd.m:103-110)"""
    if shape == "full":
        return np.arange(-(N - 1), M)
    if shape == "same":
        return np.arange(0, M) - (N - 1) // 2
    if shape == "valid":
        return np.arange(0, M - N + 1)
    if shape == "causal":                                    # extension: the first M outputs of 'full' (MATLAB filter(b, 1, x))
        return np.arange(-(N - 1), M - (N - 1))
    raise ValueError("shape must be one of {'full', 'same', 'valid'}")


def _first_nonsingleton(*arrs):
    ds = [next((k for k, v in enumerate(a.shape) if v != 1), None) for a in arrs]
    ds = [d for d in ds if d is not None]
    return (min(ds) if ds else 0) + 1


def convd(x, y=None, dim: int | None = None, shape: str = "full", device=None, return_lags: bool = False):
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("qups_amd: no HIP device visible -- convd has no CPU fallback")
    if shape not in _SHAPES:
        raise ValueError("shape must be one of {'full', 'same', 'valid'}")
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    as_t = lambda a: a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    xt = as_t(x)
    if dim is None:
        dim = _first_nonsingleton(xt, as_t(y)) if y is not None else _first_nonsingleton(xt)
    if dim < 1:
        raise ValueError("dim must be a positive integer")
    d = dim - 1
    yt = as_t(y) if y is not None else None
    D = max(xt.ndim, yt.ndim if yt is not None else 0, dim)
    xt = xt.reshape(tuple(xt.shape) + (1,) * (D - xt.ndim))
    if yt is None:
        yt = torch.conj(torch.flip(xt, (d,))).resolve_conj()
    yt = yt.reshape(tuple(yt.shape) + (1,) * (D - yt.ndim))
    for t in (xt, yt):
        if not (t.is_floating_point() or t.is_complex()):
            raise TypeError("convd expects floating-point data")
    # computation type (kern/convd.m:259-268): half if either operand is half (convh / convch, src/convd.cu:141,153), else single if either is
    # single, else double; complex if either is complex
    half = any(t.dtype in (torch.float16, torch.complex32) for t in (xt, yt))
    single = any(t.dtype in (torch.float32, torch.complex64) for t in (xt, yt))
    cplx = xt.is_complex() or yt.is_complex()
    if half:
        dt = torch.complex32 if cplx else torch.float16
    else:
        dt = {(True, True): torch.complex64, (True, False): torch.float32, (False, True): torch.complex128, (False, False): torch.float64}[(single, cplx)]
    y_real = cplx and xt.is_complex() and not yt.is_complex()      # real taps on complex data: kept real (half the multiplies; same numbers)
    rdt = {torch.complex32: torch.float16, torch.complex64: torch.float32, torch.complex128: torch.float64}.get(dt)
    sx, sy = list(xt.shape), list(yt.shape)
    other = [k for k in range(D) if k != d]
    if not all(sx[k] == sy[k] or sx[k] == 1 or sy[k] == 1 for k in other):
        raise ValueError(f"Incompatible sizes {sx}, and {sy}.")
    full = [max(sx[k], sy[k]) if k != d else 0 for k in range(D)]
    M, N = sx[d], sy[d]
    S = int(np.prod(full[:d])) if d else 1                      # slow dimensions (in front of dim in row-major order)
    Cc = int(np.prod(full[d + 1:])) if d + 1 < D else 1         # fast dimensions
    bits = 0

    def prep(t, sz, one_col, one_slice, to=None):
        nonlocal bits
        to = to or dt
        lead, trail = sz[:d], sz[d + 1:]
        lead_ok = lead == full[:d] or all(v == 1 for v in lead)
        trail_ok = trail == full[d + 1:] or all(v == 1 for v in trail)
        if not (lead_ok and trail_ok):                          # partial broadcast: replicate (the reference always does)
            tgt = list(full); tgt[d] = sz[d]
            t = t.expand(tgt)
            lead, trail = full[:d], full[d + 1:]
        if Cc > 1 and all(v == 1 for v in trail):
            bits |= one_col
        if S > 1 and all(v == 1 for v in lead):
            bits |= one_slice
        t = t.to(dev)
        if to == torch.complex32 and t.dtype != torch.complex32:             # (torch has no direct cast to complex half)
            t = torch.complex(t.real.to(torch.float32), t.imag.to(torch.float32)) if t.is_complex() else torch.complex(t.to(torch.float32), torch.zeros_like(t, dtype=torch.float32))
            t = torch.view_as_complex(torch.view_as_real(t).to(torch.float16).contiguous())
        return t.to(dtype=to).contiguous()

    xd = prep(xt, sx, _lib.QDAS_CONV_X_ONE_COLUMN, _lib.QDAS_CONV_X_ONE_SLICE)
    yd = prep(yt, sy, _lib.QDAS_CONV_Y_ONE_COLUMN, _lib.QDAS_CONV_Y_ONE_SLICE, rdt if y_real else None)
    lags = conv_lags(M, N, shape)
    L = len(lags)
    osz = list(full); osz[d] = L
    z = torch.empty(osz, dtype=dt, device=dev)
    if z.numel():
        desc = _lib.ConvdDesc(Cc, M, N, S, _lib.QDAS_F16 if half else (_lib.QDAS_F32 if single else _lib.QDAS_F64), int(cplx), _SHAPES[shape], bits,
                              dev.index if dev.index is not None else torch.cuda.current_device(), int(y_real))
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().qdas_convd(C.byref(desc), C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()), C.c_void_p(z.data_ptr()),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    if return_lags:
        lsz = [1] * D; lsz[d] = L
        return z, lags.reshape(lsz)
    return z


def sosfilt(x, sos, dim: int = 1, gain: float = 1.0, device=None):
    """Recursive (IIR) filtering along ``dim`` (1-based) with second-order sections ``sos`` (``n x 6``: ``[b0 b1 b2 a0 a1 a2]`` per row -- MATLAB's
    ``digitalFilter.Coefficients`` / ``scipy.signal`` ``sos``) and an overall ``gain``: what ``filter(D, x)`` computes for an IIR ``digitalFilter``
    (reference ``src/ChannelData.m:857-888``).  float32 / float64, real or complex, on the device (``qdas_iir``, ``csrc/iir.hip``: one read and one write of
    the record whatever the order); no CPU fallback.  Returns a tensor of ``x``'s shape."""
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("qups_amd: no HIP device visible -- sosfilt has no CPU fallback")
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    xt = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
    if not (xt.is_floating_point() or xt.is_complex()):
        xt = xt.to(torch.float64)
    if xt.dtype in (torch.float16, torch.complex32, torch.bfloat16):
        raise ValueError("sosfilt: double or single data")
    s = np.ascontiguousarray(np.asarray(sos, dtype=np.float64).reshape(-1, 6))
    if not (1 <= s.shape[0] <= 16):
        raise ValueError("sosfilt: 1 to 16 second-order sections")
    ax = int(dim) - 1
    if float(gain) == 0.0:                                        # (the C descriptor reads gain == 0 as "not set" = 1, include/qdas.h; an EXPLICIT zero gain is zero output)
        return torch.zeros_like(xt.to(dev))
    xm = xt.to(dev).movedim(ax, -1).contiguous()                  # (..., T): time fastest = the ABI's T x K column-major
    T = int(xm.shape[-1])
    K = int(xm.numel() // T) if T else 0
    y = torch.empty_like(xm)
    dbl = xm.dtype in (torch.float64, torch.complex128)
    d = _lib.IirDesc(T, K, int(s.shape[0]), _lib.QDAS_F64 if dbl else _lib.QDAS_F32, int(xm.is_complex()),
                     dev.index if dev.index is not None else torch.cuda.current_device(), float(gain), s.ctypes.data_as(C.POINTER(C.c_double)))
    L = _lib.lib()
    L.qdas_iir.argtypes = [C.POINTER(_lib.IirDesc), C.c_void_p, C.c_void_p, C.c_void_p]
    with torch.cuda.device(dev):
        _lib.check(L.qdas_iir(C.byref(d), C.c_void_p(xm.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return y.movedim(-1, ax)
