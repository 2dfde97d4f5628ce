"""Split-delay ("look-up table") sampling: the Python mirror of the reference's
``ChannelData.sample2sep`` -> ``wsinterpd2`` path that ``bfDAS`` / ``bfDASLUT`` / ``bfEikonal`` / ``focusTx``
use (reference ``src/ChannelData.m:1338-1447``, ``kern/wsinterpd2.m:1-320``; device kernel
``src/interpd.cu:344-396``).  The compute is ``qdas_das_lut`` of ``libqdas.so`` (``include/qdas.h``).

    y[i,(n),(m),f] = sum  w[i,n,m] * exp(omega * s) * sample(x[:,n,m,f], s),     s = t_rx[i,n] + t_tx[i,m]

Differences from the reference by design: outputs are owned by one lane and summed in a fixed order
(deterministic; the reference uses float atomics, ``src/interpd.cu:339,393``), and out-of-support samples
are exactly 0 (the device's ``no_v``; the ``extrapval`` argument of the MATLAB branch is accepted only as 0/NaN).
"""
from __future__ import annotations

import ctypes as C

import os

import numpy as np

from . import _lib
from .das_spec import DasError, _cast_data, _colmajor, _is_torch, _PREC


def _torch():
    import torch
    return torch


def _real_dtype(prec):
    torch = _torch()
    return torch.float64 if prec == "double" else torch.float32


def das_lut(x, tau_rx, tau_tx, *, interp="linear", w=None, keep_rx=False, keep_tx=False, omega=0.0, prec=None,
            tpose=False, device=None):
    """Weighted, phase-rotated sampling with separable delays (sample units, 0-based: ``x(1)`` in MATLAB
    is ``t == 0``, reference ``kern/wsinterpd2.m:20``).

    ``x``: ``T x N x M x F...`` (``T x M x N`` with ``tpose``); ``tau_rx``: ``I... x N``; ``tau_tx``: ``I... x M``
    (same leading pixel dims); ``w``: broadcastable to ``I... x N x M`` (real or complex) or ``None``;
    ``omega``: the IMAGINARY part of the reference's ``omega = 2i*pi*fmod/fs`` (``src/ChannelData.m:1439``).
    Returns ``I... x [1|N] x [1|M] x F...``.
    """
    torch = _torch()
    L = _lib.lib()
    if not torch.cuda.is_available():
        raise RuntimeError("qups_amd: no HIP device visible -- the sampling path has no CPU fallback")
    if interp not in _lib.INTERP_FLAGS:
        raise DasError("Interp option not recognized: " + str(interp))
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    if prec is None:
        from .das_spec import _default_prec
        prec = _default_prec(x)
    xd = _cast_data(x, prec, dev)
    while xd.ndim < 3:
        xd = xd.unsqueeze(-1)
    T = xd.shape[0]
    N, M = (xd.shape[2], xd.shape[1]) if tpose else (xd.shape[1], xd.shape[2])
    fsz = tuple(xd.shape[3:])
    F = int(np.prod(fsz)) if fsz else 1
    rt = _real_dtype(prec)
    as_t = lambda a: (a if _is_torch(a) else torch.from_numpy(np.asarray(a))).to(dev)
    trx, ttx = as_t(tau_rx).to(rt), as_t(tau_tx).to(rt)
    if trx.shape[-1] != N or ttx.shape[-1] != M or trx.shape[:-1] != ttx.shape[:-1]:
        raise DasError("Delay tables must be I... x N and I... x M with identical pixel dimensions.",
                       "QUPS:UltrasoundSystem:bfDASLUT:incompatibleReceiveDelayTable")
    Isz = tuple(trx.shape[:-1])
    I = int(np.prod(Isz)) if Isz else 1
    trx_c = _colmajor(trx.reshape(I, N)) if len(Isz) <= 1 else _colmajor(trx).reshape(N, I)
    ttx_c = _colmajor(ttx.reshape(I, M)) if len(Isz) <= 1 else _colmajor(ttx).reshape(M, I)
    d = _lib.LutDesc()
    d.T, d.N, d.M, d.I = T, N, M, I
    d.I1 = int(Isz[0]) if len(Isz) >= 2 else 0                      # fastest pixel dimension: lets the fused tiled kernel take the call
    d.flag = _lib.INTERP_FLAGS[interp] + 8 * bool(keep_rx) + 16 * bool(keep_tx) + 32 * bool(tpose)
    d.dtype = _PREC[prec]
    d.omega = float(omega)
    d.tau_rx, d.tau_tx = trx_c.data_ptr(), ttx_c.data_ptr()
    wc = None
    if w is not None:
        wt = as_t(w)
        shp = tuple(wt.shape) + (1,) * (len(Isz) + 2 - wt.ndim)
        wt = wt.reshape(shp)
        full = Isz + (N, M)
        if len(shp) != len(full) or any(s not in (1, f) for s, f in zip(shp, full)):
            raise DasError("The weighting vector w must have dimensions compatible with the data.")
        # pixel dims must be all-singleton or all-full so that one stride describes them
        pix = shp[:len(Isz)]
        if not (all(s == 1 for s in pix) or tuple(pix) == Isz):
            wt = wt.expand(Isz + shp[len(Isz):])
            shp = tuple(wt.shape)
            pix = Isz
        wI = int(np.prod(pix)) if pix else 1
        real = not wt.is_complex()
        if prec == "halfT":
            wt = wt.to(torch.float16) if real else torch.view_as_complex(torch.view_as_real(wt.to(torch.complex64)).to(torch.float16).contiguous())
        else:
            wt = wt.to(rt if real else (torch.complex128 if prec == "double" else torch.complex64))
        wc = _colmajor(wt).contiguous()
        st, acc = [], 1
        for s in (wI, shp[-2], shp[-1]):
            st.append(0 if s == 1 else acc)
            acc *= s
        d.wstride = (C.c_uint64 * 3)(*st)
        d.w, d.w_real = wc.data_ptr(), int(real)
    oN, oM = (N if keep_rx else 1), (M if keep_tx else 1)
    xc = _colmajor(xd)                                   # (F.., M, N, T)
    from .das_spec import _data_dtype
    y = torch.empty((F, oM, oN, I), dtype=_data_dtype(prec), device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    es = xc.element_size()
    with torch.cuda.device(dev):
        for f in range(F):
            _lib.check(L.qdas_das_lut(C.byref(d), C.c_void_p(xc.data_ptr() + f * T * N * M * es),
                                      C.c_void_p(y.data_ptr() + f * oM * oN * I * es), stream))
    buf = C.create_string_buffer(200)
    L.qdas_das_lut_last_kernel.argtypes = [C.c_char_p, C.c_size_t]
    if L.qdas_das_lut_last_kernel(buf, 200) == 0:
        das_lut.last_kernel = buf.value.decode()         # which kernel served the last frame (tests / tools): "tiled,mirror ... [jit <key>]" | "tiled" | "generic"
    rev = lambda t: t.permute(*reversed(range(t.ndim)))
    return rev(y.reshape(tuple(reversed(fsz)) + (oM, oN) + tuple(reversed(Isz))))


_SHIFT_MEMO: "collections.OrderedDict" = None          # device copies of the last few (shift, w) tables handed over as HOST arrays
_SHIFT_MEMO_LOCK = __import__("threading").Lock()      # (focusTx from several host threads)


def _shift_tables(shift, w, M, dbl, cplx_data, dev, torch):
    """``(shift, w, w_real)`` on the device in the kernel's layout (``m`` fastest).  Host (numpy) tables -- what ``focusTx`` computes from the sequence on
    every call -- are remembered by content: a pageable upload blocks the host until the stream has drained (~0.1 ms each at C1, round 6), and a stream
    of frames synthesises the same transmits again and again."""
    global _SHIFT_MEMO
    rt = torch.float64 if dbl else torch.float32
    host = not _is_torch(shift) and (w is None or not _is_torch(w))
    key = None
    if host:
        import collections, hashlib
        sa = np.ascontiguousarray(np.asarray(shift, np.float64))
        wa = None if w is None else np.ascontiguousarray(np.asarray(w))
        h = hashlib.blake2b(sa.tobytes(), digest_size=16)
        if wa is not None:
            h.update(str((wa.dtype.str, wa.shape)).encode()); h.update(wa.tobytes())
        key = (h.digest(), sa.shape, bool(dbl), bool(cplx_data), str(dev))
        with _SHIFT_MEMO_LOCK:
            if _SHIFT_MEMO is None:
                _SHIFT_MEMO = collections.OrderedDict()
            hit = _SHIFT_MEMO.get(key)
            if hit is not None:
                _SHIFT_MEMO.move_to_end(key)
                return hit
    sh = (shift if _is_torch(shift) else torch.from_numpy(np.asarray(shift, np.float64))).to(dev)
    if sh.ndim != 2 or sh.shape[0] != M:
        raise DasError("shift_sum: shift must be M x Mo")
    Mo = int(sh.shape[1])
    shc = sh.to(rt).t().contiguous()                                  # memory: m fastest
    wc, w_real = None, 1
    if w is not None:
        wt = (w if _is_torch(w) else torch.from_numpy(np.array(w))).to(dev)
        wt = torch.broadcast_to(wt, (M, Mo))
        if wt.is_complex():
            if not cplx_data:
                raise DasError("shift_sum: real data take real weights")
            wc, w_real = wt.to(torch.complex128 if dbl else torch.complex64).t().contiguous(), 0
        else:
            wc = wt.to(rt).t().contiguous()
    out = (shc, wc, w_real)
    if key is not None:
        torch.cuda.current_stream(dev).synchronize()                  # (once per table: a later call may use the copies from another stream)
        with _SHIFT_MEMO_LOCK:
            _SHIFT_MEMO[key] = out
            while len(_SHIFT_MEMO) > 8:
                _SHIFT_MEMO.popitem(last=False)
    return out


def shift_sum(x, shift, w=None, interp="linear", To=None, device=None, tpad=0):
    """Transmit synthesis: ``y[t', n, m', f] = sum_m w[m, m'] * x(t' + shift[m, m'], n, m, f)`` for ``t' = 0 .. To-1`` (``qdas_shift_sum``,
    ``qups_amd/csrc/shiftsum.hip``) -- what ``UltrasoundSystem.focusTx`` asks of ``sample2sep`` (reference ``src/UltrasoundSystem.m:3498``) with the
    positions written as the record's time grid plus one offset per (element, synthesised transmit).

    ``x``: ``T x N x M x F...`` float32 / float64 / complex64 / complex128; ``shift``: ``M x Mo`` in samples; ``w``: ``M x Mo`` (real, or complex for
    complex data) or ``None``.  ``tpad``: the record counts as followed by ``tpad`` zero samples (``ChannelData.zeropad`` without the copy: a tap in the
    tail is an in-range zero, ``include/qdas.h``).  Returns ``To x N x Mo x F...`` on the device (``To`` defaults to ``T + tpad``)."""
    torch = _torch()
    L = _lib.lib()
    if not torch.cuda.is_available():
        raise RuntimeError("qups_amd: no HIP device visible -- the sampling path has no CPU fallback")
    if interp not in _lib.INTERP_FLAGS:
        raise DasError("Interp option not recognized: " + str(interp))
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    xt = (x if _is_torch(x) else torch.from_numpy(np.asarray(x))).to(dev)
    if xt.dtype not in (torch.float32, torch.float64, torch.complex64, torch.complex128):
        raise DasError("shift_sum: single or double precision data")
    while xt.ndim < 3:
        xt = xt.unsqueeze(-1)
    T, N, M = (int(v) for v in xt.shape[:3])
    fsz = tuple(int(v) for v in xt.shape[3:])
    F = int(np.prod(fsz)) if fsz else 1
    dbl = xt.dtype in (torch.float64, torch.complex128)
    rt = torch.float64 if dbl else torch.float32
    shc, wc, w_real = _shift_tables(shift, w, M, dbl, xt.is_complex(), dev, torch)
    Mo = int(shc.shape[0])
    tpad = int(tpad)
    if tpad < 0:
        raise DasError("shift_sum: tpad must not be negative")
    To = T + tpad if To is None else int(To)
    xc = _colmajor(xt.reshape((T, N, M) if F == 1 else (T, N, M, F)).contiguous())    # (F, M, N, T); three dimensions take the LDS-tiled transpose
    y = torch.empty((F, Mo, N, To), dtype=xt.dtype, device=dev)
    d = _lib.ShiftDesc(T + tpad, To, N, M, Mo, F, _lib.INTERP_FLAGS[interp], 0 if dbl else 1, int(xt.is_complex()), w_real,
                       dev.index if dev.index is not None else torch.cuda.current_device(), tpad, shc.data_ptr(), wc.data_ptr() if wc is not None else None)
    with torch.cuda.device(dev):
        _lib.check(L.qdas_shift_sum(C.byref(d), C.c_void_p(xc.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return y.permute(3, 2, 1, 0).reshape((To, N, Mo) + fsz)


def sample2sep(x, t0, fs, tau1, tau2, interp="linear", w=None, sdim=(), fmod=0.0, **kw):
    """``ChannelData.sample2sep`` for data ordered ``T x N x M x F...`` (reference ``src/ChannelData.m:1338-1447``):
    ``tau1`` (``I... x N``, receive) and ``tau2`` (``I... x M``, transmit) are TIMES; the sample delays are
    ``(tau - t0) * fs`` (``:1431-1435``) and the phasor is ``exp(2i*pi*fmod/fs * ntau)`` (``:1439``).
    ``sdim`` lists the aperture dims to sum, counted like the reference's default ``apdim``: ``'rx'`` / ``'tx'``."""
    t0a = np.asarray(t0, dtype=np.float64).reshape(-1)
    torch = _torch()
    as_t = lambda a: a if _is_torch(a) else torch.from_numpy(np.asarray(a))
    t1, t2 = as_t(tau1).to(torch.float64), as_t(tau2).to(torch.float64)
    M = t2.shape[-1]
    if t0a.size not in (1, M):
        raise DasError("t0 must be a scalar or have one value per transmit.")
    t0t = torch.from_numpy(np.broadcast_to(t0a, (M,)).copy()).to(t2.device)
    n1, n2 = t1 * fs, (t2 - t0t) * fs
    sdim = set(sdim)
    return das_lut(x, n1, n2, interp=interp, w=w, keep_rx="rx" not in sdim, keep_tx="tx" not in sdim,
                   omega=2 * np.pi * fmod / fs, **kw)


def wsinterpd2(x, t1, t2, dim=1, w=1, sdim=None, interp="linear", extrapval=0, omega=0, **kw):
    """``y = wsinterpd2(x, t1, t2, dim, w, sdim, interp, extrapval, omega)`` (reference ``kern/wsinterpd2.m:1-47``) for
    the separable layouts the beamformers use: after moving ``dim`` first, ``x`` is ``T x N x M x F...`` (``N``/``M``
    may be 1), and each of ``t1``, ``t2`` (``I x [1|N] x [1|M]``) depends on the pixel and on AT MOST ONE of the two
    aperture dims.  ``sdim`` (1-based, as in MATLAB) may contain 2 and/or 3.  ``omega`` is complex as in the reference
    (only its imaginary part rotates the phase; a real part is not supported on the device either,
    ``kern/wsinterpd2.m:102``)."""
    torch = _torch()
    if not (extrapval == 0 or (isinstance(extrapval, float) and np.isnan(extrapval))):
        raise DasError("Only extrapval 0 (the device semantics) is supported.")
    if np.real(omega) != 0:
        raise DasError("omega must be purely imaginary on the device path.")
    as_t = lambda a: a if _is_torch(a) else torch.from_numpy(np.asarray(a))
    mv = lambda a: as_t(a).movedim(dim - 1, 0) if as_t(a).ndim >= dim else as_t(a)
    x, t1, t2 = mv(x), mv(t1), mv(t2)
    pad3 = lambda a: a.reshape(tuple(a.shape) + (1,) * max(0, 3 - a.ndim))
    x, t1, t2 = pad3(x), pad3(t1), pad3(t2)
    if not (torch.is_floating_point(t1) and torch.is_floating_point(t2)):
        raise DasError("Sample indices must be real.")
    N = max(x.shape[1], t1.shape[1], t2.shape[1])
    M = max(x.shape[2], t1.shape[2], t2.shape[2])
    I = max(t1.shape[0], t2.shape[0])
    rx = torch.zeros((I, N), dtype=torch.float64)
    tx = torch.zeros((I, M), dtype=torch.float64)
    for t in (t1, t2):
        if t.ndim > 3 and any(s != 1 for s in t.shape[3:]):
            raise DasError("Delays may only vary over the pixel and aperture dimensions.")
        t = t.reshape(t.shape[:3]).to(torch.float64).cpu()
        if t.shape[1] > 1 and t.shape[2] > 1:
            raise DasError("Delays must be separable: each of t1, t2 may depend on at most one aperture dimension.")
        if t.shape[2] > 1:
            tx = tx + t[:, 0, :].expand(I, M)
        else:
            rx = rx + t[:, :, 0].expand(I, N)
    xb = x.expand((x.shape[0], N, M) + tuple(x.shape[3:]))
    sd = set(np.atleast_1d(sdim).astype(int).tolist()) if sdim is not None and np.size(sdim) else set()
    if not sd <= {2, 3}:
        raise DasError("Only the aperture dimensions (2, 3) can be summed on the device path.")
    wt = None if (np.isscalar(w) and w == 1) else pad3(mv(w))
    y = das_lut(xb, rx, tx, interp=interp, w=wt, keep_rx=2 not in sd, keep_tx=3 not in sd, omega=float(np.imag(omega)), **kw)
    return y.movedim(0, dim - 1) if dim != 1 else y


# ------------------------------------------------------------------------------------------
# General single-delay flavour (reference kern/wsinterpd.m, kern/interpd.m): one delay array t that may depend on every
# dimension; matching / outer dimensions of x and t broadcast against each other.  Runs on qdas_wsinterpd (csrc/wsinterpd.hip).
# ------------------------------------------------------------------------------------------
def _pad_shape(shape, nd):
    return tuple(shape) + (1,) * (nd - len(shape))


def _col_strides(shape):
    """element strides of a column-major array, 0 on singleton dimensions (kern/wsinterpd.m:106-116)"""
    st, acc = [], 1
    for s in shape:
        st.append(0 if s == 1 else acc)
        acc *= s
    return st


def wsinterpd(x, t, dim=1, w=1, sdim=None, interp="linear", extrapval=float("nan"), omega=0, prec=None, device=None):
    """``y = wsinterpd(x, t, dim, w, sdim, interp, extrapval, omega)`` -- reference ``kern/wsinterpd.m:1-43``.

    ``x`` is ``T x N' x F'``, ``t`` is ``I x N' x M'`` (sample indices, 0-based: ``x(1)`` in MATLAB is ``t == 0``) after swapping
    ``dim`` (1-based) with dimension 1; matching dimensions are sampled element-wise, dimensions that are singleton in one of the
    two broadcast (``:70-93``); ``w`` (sizes 1 or full, ``:84-88``) multiplies the samples, ``exp(omega .* t)`` rotates them
    (``omega`` purely imaginary on the device, ``:102``), dimensions ``sdim`` are summed.  Result ``I x N' x M' x F'`` with the
    summed dimensions singleton, ``dim`` swapped back.  ``extrapval`` is the value of samples outside the record (``NaN`` by
    default as in the reference; sums omit NaN, ``:262``)."""
    torch = _torch()
    L = _lib.lib()
    if not torch.cuda.is_available():
        raise RuntimeError("qups_amd: no HIP device visible -- the sampling path has no CPU fallback")
    if interp not in _lib.INTERP_FLAGS:
        raise DasError("Interp option not recognized: " + str(interp))
    if np.real(omega) != 0:
        raise DasError("omega must be purely imaginary on the device path.")
    if np.ndim(extrapval) != 0 or np.iscomplexobj(extrapval):
        raise DasError("Only a scalar value accepted for extrapolation.")
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    as_t = lambda a: a if _is_torch(a) else torch.from_numpy(np.asarray(a))
    xt, tt = as_t(x), as_t(t)
    if tt.is_complex():
        raise DasError("Sample indices must be real.")
    if prec is None:
        from .das_spec import _default_prec
        prec = _default_prec(xt)
    wscalar = not _is_torch(w) and np.ndim(w) == 0
    wt = None if (wscalar and w == 1) else as_t(np.asarray(w) if not _is_torch(w) else w)
    nd = max(xt.ndim, tt.ndim, wt.ndim if wt is not None else 0, dim)
    if nd > 8:
        raise DasError("At most 8 dimensions are supported.")
    pad = lambda a: a.reshape(_pad_shape(a.shape, nd))
    sw = lambda a: a.transpose(0, dim - 1) if dim != 1 else a          # swapdim(a, dim, 1)
    xt, tt = sw(pad(xt)), sw(pad(tt))
    if wt is not None:
        wt = sw(pad(wt))
    sd = [] if sdim is None else [int(v) for v in np.atleast_1d(sdim)]
    sd = [dim if v == 1 else (1 if v == dim else v) for v in sd]        # the same swap for the summed dimensions
    if any(v < 1 for v in sd):
        raise DasError("Summation dimensions must be positive.")
    # summation over dimensions that are singleton in x and t is a no-op (kern/wsinterpd.m:73)
    sd = [v for v in sd if v <= nd and not (xt.shape[v - 1] == 1 and tt.shape[v - 1] == 1)]
    T, I = xt.shape[0], tt.shape[0]
    size = [I]
    for d in range(1, nd):
        xs, ts = xt.shape[d], tt.shape[d]
        if xs != ts and xs != 1 and ts != 1:
            raise DasError(f"Delay size must match the data size ({xs}) or be singleton in dimension {d + 1}.")
        size.append(max(xs, ts))
    if wt is not None and (wt.shape[0] not in (1, I) or any(ws not in (1, fs_) for ws, fs_ in zip(wt.shape[1:], size[1:]))):
        raise DasError("The weighting vector w must have dimensions compatible with the data and may not broadcast.")
    xd = _cast_data(xt, prec, dev)
    rt = _real_dtype(prec)
    td = tt.to(dev).to(rt)
    # x and t are sampled IN PLACE, whatever their memory order (round 3 made a column-major copy of both first -- a read + write pass over the
    # record that cost as much as the sampling itself): the kernel takes element strides per dimension, y is laid out like x (dimension order of
    # x's memory) and the lanes of a wave run along the dimension in which x is contiguous.
    def _dense(a):
        """strides usable as they are: no two elements of `a` alias (expanded / overlapping views are copied)"""
        if a.numel() == 0:
            return True
        sz_st = sorted(((a.shape[k], abs(a.stride(k))) for k in range(a.ndim) if a.shape[k] > 1), key=lambda v: v[1])
        need = 1
        for n, st in sz_st:
            if st < need:
                return False
            need = st * (n - 1) + need
        return all(a.stride(k) >= 0 for k in range(a.ndim))
    if not _dense(xd):
        xd = xd.contiguous()
    if not _dense(td):
        td = td.contiguous()
    # ... unless the dimension in which x is contiguous is SUMMED: a lane then walks its own row term by term and neighbouring lanes sit a whole
    # row apart -- the column-major copies (time fastest: lanes along the sampled dimension, sums by uniform strides) are the better layout there
    nz = [k for k in range(1, nd) if xd.shape[k] > 1]
    fastest = min(nz, key=lambda k: xd.stride(k)) if nz else 0
    # (round 6: ONE summed dimension of >= 16 terms, unit stride: the library sums across the lanes of a wave -- csrc/wsinterpd.hip
    #  wsinterpd_lanesum_kernel -- and the record is used where it lies; QDAS_WS_NO_LANESUM: the transposed form, for A/B runs)
    lanesum = (len([k for k in sd if xd.shape[k - 1] > 1 or td.shape[k - 1] > 1]) == 1 and nz and xd.stride(fastest) == 1 and max(xd.shape[fastest], td.shape[fastest]) >= 16
               and not os.environ.get("QDAS_WS_NO_LANESUM"))
    if nz and T > 1 and xd.stride(fastest) < xd.stride(0) and (fastest + 1) in sd and not lanesum:
        xd, td = _colmajor(xd).permute(*reversed(range(nd))), _colmajor(td).permute(*reversed(range(nd)))
    xc, tc = xd, td
    d = _lib.WsDesc()
    d.T, d.x_tstride, d.ndim, d.flag, d.dtype = T, (xd.stride(0) if T > 1 else 1), nd, _lib.INTERP_FLAGS[interp], _PREC[prec]
    xs_ = [0 if xd.shape[k] == 1 else xd.stride(k) for k in range(nd)]
    xs_[0] = 0
    ts_ = [0 if td.shape[k] == 1 else td.stride(k) for k in range(nd)]
    for k in range(nd):
        d.size[k], d.tstride[k], d.xstride[k] = size[k], ts_[k], xs_[k]
        d.sum[k] = 1 if (k + 1) in sd else 0
    wc = None
    if wt is not None:
        real = not wt.is_complex()
        wt = wt.to(dev)
        if prec == "halfT":
            wt = wt.to(torch.float16) if real else torch.view_as_complex(torch.view_as_real(wt.to(torch.complex64)).to(torch.float16).contiguous())
        else:
            wt = wt.to(rt if real else (torch.complex128 if prec == "double" else torch.complex64))
        wc = _colmajor(wt).contiguous()
        for k, v in enumerate(_col_strides(wt.shape)):
            d.wstride[k] = v
        d.w, d.w_real = wc.data_ptr(), int(real)
    d.omega, d.extrap = float(np.imag(omega)), float(extrapval)
    d.t, d.x = tc.data_ptr(), xc.data_ptr()
    osz = [1 if (k + 1) in sd else size[k] for k in range(nd)]
    from .das_spec import _data_dtype
    # y: the dimension order of x's memory (fastest first: smallest x stride; dimensions x broadcasts over ordered by t's stride, then by index);
    # dimension 0 -- the sampled one -- takes the place of x's time dimension
    key = lambda k: (xd.stride(0) if k == 0 else xs_[k]) if (k == 0 or xs_[k]) else (1 << 62) + (ts_[k] if ts_[k] else (1 << 61) + k)
    order = sorted(range(nd), key=key)
    yst, acc = [0] * nd, 1
    for k in order:
        yst[k] = acc
        acc *= osz[k]
    y = torch.empty_strided(tuple(osz), tuple(yst), dtype=_data_dtype(prec), device=dev)
    kept = [k for k in order if osz[k] > 1 and (k + 1) not in sd]
    for k in range(nd):
        d.ystride[k] = yst[k]
    d.lane_dim = kept[0] if kept else -1
    with torch.cuda.device(dev):
        _lib.check(L.qdas_wsinterpd(C.byref(d), C.c_void_p(y.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return sw(y)


def interpd(x, t, dim=1, interp="linear", extrapval=float("nan"), **kw):
    """``y = interpd(x, t, dim, interp, extrapval)`` -- reference ``kern/interpd.m:1-47``: ``wsinterpd`` without weights or sums."""
    return wsinterpd(x, t, dim, 1, None, interp, extrapval, 0, **kw)
