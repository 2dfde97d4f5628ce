"""Pre-processing in front of the DAS path, on the device: ``hilbert`` (+ fused ``downmix``) of real RF traces.

Mirrors ``ChannelData.hilbert`` (reference src/ChannelData.m:935-966) and ``ChannelData.downmix`` (:757-766) for data whose time
axis is the first dimension.  The work is done by ``libqdas.so`` (qups_amd/csrc/pre.hip: a one-pass LDS-resident FFT kernel for
record lengths with prime factors up to 13, hipFFT passes for the rest); there is no CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


# Plans by shape: a qdas_pre_plan owns its twiddles -- and, for record lengths the one-pass kernel does not take (a prime factor above 13, more than 8192
# samples), two hipFFT plans and three work buffers whose creation costs 11-15 ms (profiles/r04/general_time.txt) against 0.1-0.7 ms of transform.  A frame
# loop calls hilbert() with ONE shape: the last few plans are kept, destroyed on eviction and at interpreter exit.
_PLANS: "dict[tuple, tuple]" = {}           # key -> (plan handle, the plan's own lock)
_PLAN_CACHE_MAX = 8
_plans_lock = __import__("threading").Lock()      # guards the DICTIONARY only (it used to be held across execute + stream synchronisation: every hilbert() of every device and thread in a row)


def _destroy(h, lk):
    with lk:                                      # (never under a running execute)
        try:
            _lib.lib().qdas_pre_plan_destroy(h)
        except Exception:                         # (interpreter exit: the HIP runtime may be gone before this hook runs)
            pass


def clear_pre_plan_cache():
    with _plans_lock:
        items = list(_PLANS.values())
        _PLANS.clear()
    for h, lk in items:
        _destroy(h, lk)


def _at_exit():
    try:
        clear_pre_plan_cache()
    except Exception:
        pass


__import__("atexit").register(_at_exit)


def _pre_plan(key, d):
    """(plan, lock) of this shape: created on a miss, the oldest entry evicted beyond _PLAN_CACHE_MAX (destroyed once no execute holds it)"""
    L = _lib.lib()
    evicted = []
    with _plans_lock:
        e = _PLANS.pop(key, None)
        if e is None:
            h = C.c_void_p()
            _lib.check(L.qdas_pre_plan_create(C.byref(h), C.byref(d)))
            hilbert.plans_created = getattr(hilbert, "plans_created", 0) + 1
            e = (h, __import__("threading").Lock())
        _PLANS[key] = e                                              # (most recently used last)
        while len(_PLANS) > _PLAN_CACHE_MAX:
            evicted.append(_PLANS.pop(next(iter(_PLANS))))
    for h, lk in evicted:
        _destroy(h, lk)
    return e


def hilbert(x, N: int | None = None, fdown: float = 0.0, t0: float = 0.0, fs: float | None = None, device=None):
    """Analytic signal along dim 0 of real ``x`` (``T x ...``; float32 or int16; numpy array or torch tensor), transform length
    ``N`` (default ``T``; zero-padded or truncated like MATLAB's ``hilbert(x, N)``); with ``fdown`` the result is also multiplied by
    ``exp(-2j*pi*fdown*(t0 + k/fs))``.  Returns a complex64 torch tensor ``N x ...`` on the device.

    ``t0``, ``fs`` and ``fdown`` are constants of the underlying plan (``qdas_pre_desc``), so they are part of the plan cache's key: a frame loop whose ``t0`` changes
    per frame creates a plan per value (0.1 ms on the one-pass path, 11-15 ms where hipFFT plans are made) -- downmix with ``t0 = 0`` and multiply the frame by the
    scalar ``exp(-2j*pi*fdown*t0)`` instead."""
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
    key = (T, K, N, int(d.in_type), int(d.device), float(d.fs), float(d.t0), float(d.fdown), __import__("os").environ.get("QDAS_PRE_HIPFFT"))      # (the switch is read at plan creation: part of the key)
    h, plan_lock = _pre_plan(key, d)
    with torch.cuda.device(dev), plan_lock:                         # (one execute at a time per PLAN: its hipFFT work buffers are the plan's; other shapes / devices run beside it)
        hilbert.last_one_pass = bool(L.qdas_pre_plan_one_pass(h))   # which path served the last call (tests / tools)
        _lib.check(L.qdas_pre_execute(h, C.c_void_p(xc.data_ptr()), C.c_void_p(y.data_ptr()),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.current_stream().synchronize()
    return y.t().reshape((N,) + rest)
