"""Pre-processing in front of the DAS path, on the device: ``hilbert`` (+ fused ``downmix``) of real RF traces.

Mirrors ``ChannelData.hilbert`` (reference src/ChannelData.m:935-966) and ``ChannelData.downmix`` (:757-766) for data whose time
axis is the first dimension.  The work is done by ``libqdas.so`` (qups_amd/csrc/pre.hip: a one-pass LDS-resident FFT kernel for
record lengths with prime factors up to 13, hipFFT passes for the rest); there is no CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def hilbert(x, N: int | None = None, fdown: float = 0.0, t0: float = 0.0, fs: float | None = None, device=None):
    """Analytic signal along dim 0 of real ``x`` (``T x ...``; float32 or int16; numpy array or torch tensor), transform length
    ``N`` (default ``T``; zero-padded or truncated like MATLAB's ``hilbert(x, N)``); with ``fdown`` the result is also multiplied by
    ``exp(-2j*pi*fdown*(t0 + k/fs))``.  Returns a complex64 torch tensor ``N x ...`` on the device."""
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("qups_amd: no HIP device visible -- hilbert has no CPU fallback")
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    xt = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
    if xt.is_complex():
        raise ValueError("hilbert expects real data")
    if xt.dtype not in (torch.float32, torch.int16):
        xt = xt.to(torch.float32)
    T = int(xt.shape[0])
    rest = tuple(int(v) for v in xt.shape[1:])
    K = int(np.prod(rest)) if rest else 1
    N = T if N is None else int(N)
    if fdown and not fs:
        raise ValueError("Undefined sampling rate.")
    from .das_spec import _colmajor
    x2 = xt.to(dev).reshape(T, K)
    # already time-fastest (a MATLAB-ordered view; a CONTIGUOUS single trace): no copy.  A strided single trace -- hilbert(x[:, j]) of a
    # T x N tensor, x[::2] -- has K == 1 but stride(0) != 1: the library reads T consecutive elements, so it is copied like the rest.
    if T > 0 and ((K == 1 and x2.stride(0) == 1) or (K > 1 and x2.stride() == (1, T))):
        xc = x2.t()
    else:
        xc = _colmajor(x2.contiguous())                             # K x T: time fastest (MATLAB memory order of T x K)
    y = torch.empty((K, N), dtype=torch.complex64, device=dev)
    d = _lib.PreDesc(T, K, N, _lib.QDAS_PRE_I16 if xt.dtype == torch.int16 else _lib.QDAS_PRE_F32,
                     dev.index if dev.index is not None else torch.cuda.current_device(), float(fs or 0.0), float(t0), float(fdown))
    L = _lib.lib()
    h = C.c_void_p()
    with torch.cuda.device(dev):
        _lib.check(L.qdas_pre_plan_create(C.byref(h), C.byref(d)))
        hilbert.last_one_pass = bool(L.qdas_pre_plan_one_pass(h))   # which path served the last call (tests / tools)
        try:
            _lib.check(L.qdas_pre_execute(h, C.c_void_p(xc.data_ptr()), C.c_void_p(y.data_ptr()),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            torch.cuda.current_stream().synchronize()
        finally:
            L.qdas_pre_plan_destroy(h)
    return y.t().reshape((N,) + rest)
