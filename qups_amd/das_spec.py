"""Host side of the delay-and-sum path: the Python mirror of the reference's ``das_spec``.

``das_spec(fun, Pi, Pr, Pv, Nv, x, t0, fs, c, *options)`` has the reference's signature, option
strings, argument meaning and error behaviour (reference ``kern/das_spec.m:1-148,563-670``); what
differs is only what sits below it: instead of ``parallel.gpu.CUDAKernel('bf.ptx', ...)`` +
``k.feval`` (``kern/das_spec.m:279-306,372``) the arguments are marshalled into the C-ABI of
``libqdas.so`` (``include/qdas.h``) whose kernels are hand-written for MI355X.

Array convention: arrays keep MATLAB's dimension ORDER (``x`` is ``T x N x M x F...``, ``Pi`` is
``3 x I1 x I2 x I3``, outputs are ``I1 x I2 x I3 x [1|N] x [1|M] x F...``).  They may be numpy
arrays or torch tensors with any strides; the column-major buffers the kernel ABI needs are
produced here (zero-copy when the tensor already is column-major, e.g. a ``(M, N, T)`` C-contiguous
tensor viewed as ``.permute(2, 1, 0)``).

Everything up to :class:`DasProblem` is pure host logic (numpy only) and is unit-tested without
a GPU; :class:`DasPlan` needs ``libqdas.so`` and a HIP device and fails loudly otherwise.
"""
from __future__ import annotations

import collections
import ctypes as C
import os
import threading
from dataclasses import dataclass, field
from typing import Any, Sequence

import numpy as np

from . import _lib

_FUNS = ("DAS", "SYN", "MUL", "BF", "delays")
_PREC = {"double": _lib.QDAS_F64, "single": _lib.QDAS_F32, "halfT": _lib.QDAS_F16}
_REAL_NP = {"double": np.float64, "single": np.float32, "halfT": np.float32}


class DasError(ValueError):
    """Host-side argument error; ``identifier`` mirrors the MATLAB error ID where the reference has one."""

    def __init__(self, msg, identifier=None):
        super().__init__(msg)
        self.identifier = identifier


# ------------------------------------------------------------------------------------------
# option parsing (reference kern/das_spec.m:90-148)
# ------------------------------------------------------------------------------------------
def _is_torch(a) -> bool:
    return type(a).__module__.startswith("torch")


def _default_prec(x) -> str:
    """reference kern/das_spec.m:96-104: precision follows the data type, default double."""
    dt = str(getattr(x, "dtype", "")).replace("torch.", "")
    if dt in ("float32", "complex64"):
        return "single"
    if dt in ("float16", "complex32"):
        return "halfT"
    return "double"


def parse_options(x, varargin: Sequence[Any]) -> dict:
    o = dict(VS=True, DV=False, interp="linear", apod=[], prec=_default_prec(x), device=-1,
             fmod=0.0, tpose=False, rx_apod=None)
    n = 0
    nargs = len(varargin)

    def val():
        nonlocal n
        n += 1
        if n >= nargs:
            raise DasError("Unrecognized option")
        return varargin[n]

    while n < nargs:
        key = varargin[n]
        if not isinstance(key, str):
            raise DasError("Unrecognized option")
        if key == "plane-waves":
            o["VS"] = False
        elif key == "virtual-source":
            o["VS"] = True
        elif key == "diverging-waves":
            o["DV"] = True
        elif key == "focused-waves":
            o["DV"] = False
        elif key == "input-precision":
            o["prec"] = str(val())
        elif key == "device":
            o["device"] = int(val())
        elif key == "interp":
            o["interp"] = str(val())
        elif key == "apod":
            o["apod"].append(val())
        elif key == "modulation":
            o["fmod"] = float(val())
        elif key == "transpose":
            o["tpose"] = bool(val())
        elif key == "rx-apod":                      # extension: generated in the kernel (qups_amd.apodization.rx_apod_spec)
            o["rx_apod"] = val()
        else:
            raise DasError("Unrecognized option")
        n += 1
    if o["prec"] not in _PREC:
        raise DasError(f"Unrecognized input precision {o['prec']!r}: must be one of {sorted(_PREC)}")
    return o


# ------------------------------------------------------------------------------------------
# geometry normalisation (reference kern/das_spec.m:563-670)
# ------------------------------------------------------------------------------------------
def _to_numpy(a, dtype=None) -> np.ndarray:
    if _is_torch(a):
        a = a.detach().cpu().numpy()
    a = np.asarray(a)
    return a.astype(dtype) if dtype is not None else a


def _flat_colmajor(a: np.ndarray, dtype) -> np.ndarray:
    """column-major flattening of an array, cast to ``dtype`` (reference kern/das_spec.m:344-345: ``apod(:)``).  For a C-ordered array that
    is a transposing copy; numpy's is single-threaded (0.55 s for BASELINE C5's 512 x 1024 x 128 mask), torch's is not (cast first: fewer bytes)"""
    tdt = {np.float16: "float16", np.float32: "float32", np.float64: "float64", np.complex64: "complex64", np.complex128: "complex128"}.get(dtype)
    if a.size < (1 << 16) or tdt is None or not a.flags.c_contiguous or not a.flags.aligned or a.dtype.byteorder == ">" or a.dtype.kind not in "biufc":
        return a.reshape(-1, order="F").astype(dtype)
    import torch
    import warnings
    with warnings.catch_warnings():                     # (a read-only array is only read here)
        warnings.simplefilter("ignore", UserWarning)
        t = torch.from_numpy(a).to(getattr(torch, tdt))
    return t.permute(*reversed(range(t.ndim))).contiguous().reshape(-1).numpy()


def _split_separable(a: np.ndarray, N: int, M: int):
    """``a`` (5-D, real, depends on the pixel, the receiver AND the transmit) as ``(tx, rx)`` with ``tx * rx == a`` exactly in float32 --
    ``tx`` without receiver dependence, ``rx`` without transmit dependence --, or ``None``.  Per pixel the factors are the row and the
    column through the entry of largest magnitude."""
    if a.ndim != 5 or np.iscomplexobj(a) or N < 2 or M < 2 or a.shape[3] != N or a.shape[4] != M or all(d == 1 for d in a.shape[:3]) or a.size > (1 << 26):
        return None
    w = a.astype(np.float32)
    pix = w.shape[:3]
    flat = np.abs(w).reshape(pix + (N * M,))
    piv = flat.argmax(axis=-1)                                   # pixel-shaped
    n0, m0 = piv // M, piv % M
    row = np.take_along_axis(w, n0[..., None, None], axis=3)     # pix x 1 x M: the transmit-side factor
    col = np.take_along_axis(w, m0[..., None, None], axis=4)     # pix x N x 1
    pv = np.take_along_axis(row, m0[..., None, None], axis=4)    # pix x 1 x 1
    with np.errstate(divide="ignore", invalid="ignore"):
        rx = np.where(pv != 0, col / pv, np.float32(0)).astype(np.float32)
    if not np.array_equal(row * rx, w):
        return None
    return row.astype(a.dtype if a.dtype.kind == "f" else np.float32), rx.astype(a.dtype if a.dtype.kind == "f" else np.float32)


def _mod_size(P: np.ndarray) -> np.ndarray:
    """coordinates into the first dimension (reference kern/das_spec.m:591-599)"""
    if P.ndim < 2:
        P = P.reshape(-1, 1)
    if P.shape[0] <= 4:
        return P
    if P.shape[1] <= 4 and P.ndim == 2:
        return P.T
    import warnings
    warnings.warn("Input data size is ambiguous.")
    return P


def _mod_dim(P: np.ndarray) -> np.ndarray:
    """1D -> x, 2D -> (x,z), 4D -> xyz/w (reference kern/das_spec.m:656-669)"""
    d = P.shape[0]
    if d == 3:
        return P
    z = np.zeros((1,) + P.shape[1:], dtype=P.dtype)
    if d == 1:
        return np.concatenate([P, z, z], 0)
    if d == 2:
        return np.concatenate([P[:1], z, P[1:2]], 0)
    if d == 4:
        return P[:3] / P[3:4]
    raise DasError("Improper coordinate dimension.")


def _size5(a: np.ndarray):
    if a.ndim > 5:
        raise DasError("Apodization / sound speed arrays must have at most 5 dimensions (I1 x I2 x I3 x N x M).")
    return tuple(a.shape) + (1,) * (5 - a.ndim)


def _stride_row(shape5) -> list:
    """element strides with 0 on singleton dims (reference kern/das_spec.m:259-260)"""
    st, acc = [], 1
    for s in shape5:
        st.append(0 if s == 1 else acc)
        acc *= s
    return st


@dataclass
class DasProblem:
    """Everything of one ``das_spec`` call except the data: the kernel ABI arguments on the host."""
    fun: str
    prec: str
    flag: int
    VS: bool
    DV: bool
    Isz: tuple
    T: int
    N: int
    M: int
    fsz: tuple                      # frame dims of x (dims 4+)
    fs: float
    fmod: float
    Pi: np.ndarray                  # 3 x I   (column-major flat, real(prec))
    Pr: np.ndarray                  # 3 x N
    Pv: np.ndarray                  # 4 x M, row 4 = t0 (reference kern/das_spec.m:361)
    Nv: np.ndarray                  # 3 x M
    cinv: np.ndarray                # flat, real(prec)
    apod: np.ndarray | None         # flat concatenation (reference kern/das_spec.m:344-345)
    apod_real: bool
    acstride: np.ndarray            # uint64, 6*(1+S)  (reference kern/das_spec.m:257-260)
    S: int
    tpose: bool
    interp: str
    osize: tuple = field(default=(1, 1))
    rx_apod: dict | None = None     # generated receive apodization: {'kind', 'p', 'normals' (3 x N, real(prec)) | None}

    @property
    def I(self) -> int:
        return int(np.prod(self.Isz))


def build_problem(fun, Pi, Pr, Pv, Nv, xshape, t0, fs, c, opts: dict) -> DasProblem:
    """Host marshalling of one call: reference ``kern/das_spec.m:150-170,198-213,246-269,344-361``."""
    if fun not in _FUNS:
        raise DasError("Invalid beamformer.")
    prec = opts["prec"]
    rt = _REAL_NP[prec]
    ct = {"double": np.complex128, "single": np.complex64, "halfT": np.complex64}[prec]
    interp = opts["interp"]
    if interp not in _lib.INTERP_FLAGS:
        raise DasError("Unrecognized interpolation of type " + str(interp)
                       + ": must be one of {'nearest', 'linear', 'cubic', 'lanczos3'}.",
                       "QUPS:das_spec:UnrecognizedInput")
    if c is None:
        c = 1540.0
    if fs is None:
        t0v = _to_numpy(t0).reshape(-1)
        if fun == "delays":
            fs = 1.0
        elif t0v.size > 1:
            # "find the sampling frequency" from a time axis.  DEVIATION: the reference takes fs = mean(diff(t0,1,1))
            # (kern/das_spec.m:153-155), i.e. the sampling PERIOD -- its own comment says "frequency"; this mirror uses 1 / mean(diff).
            # A per-transmit t0 vector (one start time per transmit) passed WITHOUT fs would be read as a time axis here, exactly as
            # the reference does: say so instead of silently collapsing it to min(t0).
            if xshape is not None and len(xshape) >= 3 and t0v.size == xshape[1 if opts["tpose"] else 2] and t0v.size != xshape[0]:
                import warnings
                warnings.warn("das_spec: fs is omitted and t0 has one entry per transmit: it is interpreted as a TIME AXIS "
                              "(kern/das_spec.m:153-155), not as per-transmit start times -- pass fs to use it as the latter.")
            fs = 1.0 / float(np.mean(np.diff(t0v)))
            t0 = float(t0v.min())
        else:
            raise DasError("Undefined sampling rate.")
    fs = float(_to_numpy(fs).reshape(-1)[0])

    Pi = _mod_dim(_mod_size(_to_numpy(Pi)))                  # (kept in its own type: the pixel grid is the one large geometry array, and col() casts it)
    if Pi.dtype.kind not in "fiu":
        Pi = Pi.astype(np.float64)
    Pr = _mod_dim(_mod_size(_to_numpy(Pr, np.float64)))
    Pv = _mod_dim(_mod_size(_to_numpy(Pv, np.float64)))
    Nv = _mod_dim(_mod_size(_to_numpy(Nv, np.float64)))
    if Pi.ndim > 4:
        raise DasError("Pixel positions must be 3 x I1 x I2 x I3.")
    Pi = Pi.reshape(Pi.shape + (1,) * (4 - Pi.ndim))
    Isz = tuple(int(s) for s in Pi.shape[1:4])
    Pr, Pv, Nv = (p.reshape(3, -1) for p in (Pr, Pv, Nv))

    tpose = bool(opts["tpose"])
    if fun == "delays":
        T = 0
        M = max(Pv.shape[1], Nv.shape[1])
        N = Pr.shape[1]
        fsz = ()
    else:
        xs = tuple(int(s) for s in xshape) + (1,) * max(0, 3 - len(xshape))
        T, N, M = xs[0], xs[1], xs[2]
        if tpose:
            M, N = N, M                                       # reference kern/das_spec.m:251
        fsz = xs[3:]
    Mv, Mnv, Nr = Pv.shape[1], Nv.shape[1], Pr.shape[1]
    if Mv == 1:
        Pv, Mv = np.repeat(Pv, M, 1), M                      # reference kern/das_spec.m:631-633
    if Mnv == 1:
        Nv, Mnv = np.repeat(Nv, M, 1), M
    if Nr == 1 and fun != "delays":
        Pr, Nr = np.repeat(Pr, N, 1), N
    if not (Mv == Mnv and M == Mv):
        raise DasError("Inconsistent transmitter data size.")
    if N != Nr:
        raise DasError("Inconsistent receiver data size.")

    # sound speed and apodization: broadcastable I1 x I2 x I3 x N x M (reference kern/das_spec.m:636-641)
    cinv = 1.0 / _to_numpy(c, np.float64)
    cinv = cinv.reshape(_size5(cinv))
    full = Isz + (N, M)
    if any(s not in (1, f) for s, f in zip(cinv.shape[:3], full[:3])):
        raise DasError("Sound speed data size inconsistent with pixel data size")
    if cinv.shape[3] not in (1, N):
        raise DasError("Sound speed data size inconsistent with receiver data size")
    if cinv.shape[4] not in (1, M):
        raise DasError("Sound speed data size inconsistent with transmit data size")
    apods = []
    for a in opts["apod"]:
        a = _to_numpy(a)
        a = a.reshape(_size5(a))
        if any(s not in (1, f) for s, f in zip(a.shape[:3], full[:3])):
            raise DasError("Apodization data size inconsistent with pixel data size")
        if a.shape[3] not in (1, N):
            raise DasError("Apodization data size inconsistent with receiver data size")
        if a.shape[4] not in (1, M):
            raise DasError("Apodization data size inconsistent with transmit data size")
        if a.size == 1 and a.reshape(-1)[0] == 1:
            continue                                          # the reference's default {1}: a no-op
        apods.append(a)
    # an array over pixels x receivers x transmits that is an exact product of a transmit-side and a receive-side factor (the reference's
    # translating-aperture mask, src/UltrasoundSystem.m:5162: |xi - xv| <= tol & |xi - xn| <= tol) is passed as those two factors: both run
    # on the fused kernel, the single array would not (csrc/qdas_api.hip)
    if prec == "single" and len(apods) < _lib.MAX_APOD:
        for k, a in enumerate(apods):
            two = _split_separable(a, N, M)
            if two is not None:
                apods[k:k + 1] = list(two)
                break
    if len(apods) > _lib.MAX_APOD:
        raise DasError(f"At most {_lib.MAX_APOD} apodization arrays are supported.")
    apod_real = all(not np.iscomplexobj(a) for a in apods)
    table = _stride_row(cinv.shape) + [0]
    base = 0
    for a in apods:
        table += _stride_row(a.shape) + [base]
        base += a.size
    if apods:
        adt = rt if apod_real else ct
        if prec == "halfT":
            adt = np.float16 if apod_real else None
        if adt is None:   # complex half: interleaved (re, im) float16 pairs
            flat = np.concatenate([np.stack([a.real, a.imag], 0).reshape(2, -1, order="F").T.reshape(-1)
                                   for a in apods]).astype(np.float16)
        else:
            parts = [_flat_cached(a, adt) for a in apods]
            flat = parts[0] if len(parts) == 1 else np.concatenate(parts)
    else:
        flat = None

    t0v = _to_numpy(t0, np.float64).reshape(-1) if fun != "delays" else np.zeros(1)
    if t0v.size not in (1, M):
        raise DasError("t0 must be a scalar or have one value per transmit.")
    Pv4 = np.concatenate([Pv, np.broadcast_to(t0v.reshape(1, -1), (1, M))], 0)   # reference kern/das_spec.m:361

    keep_rx, keep_tx = fun in ("SYN", "BF"), fun in ("MUL", "BF")
    flag = _lib.INTERP_FLAGS[interp] + 8 * keep_rx + 16 * keep_tx + 32 * tpose     # reference kern/das_spec.m:210-213
    # 'BF' stores plane nm, i.e. the DATA's aperture order (reference src/bf.cu:100,135): I x M x N when the
    # data is transposed.  (The reference then labels that buffer [Isz N M], kern/das_spec.m:381 -- "perm(N x M)",
    # src/UltrasoundSystem.m:3357; here the array is returned with the shape it actually has.)
    osize = {"DAS": (1, 1), "SYN": (N, 1), "MUL": (1, M), "BF": (M, N) if tpose else (N, M), "delays": (N, M)}[fun]
    col = lambda A: _col_cached(A, rt)
    rxa = opts.get("rx_apod")
    if rxa is not None:
        if not isinstance(rxa, dict) or "kind" not in rxa:
            raise DasError("'rx-apod' expects the dict returned by qups_amd.apodization.rx_apod_spec")
        nrm = rxa.get("normals")
        if nrm is not None:
            nrm = np.asarray(nrm, dtype=np.float64).reshape(3, -1)
            if nrm.shape[1] != N:
                raise DasError(f"'rx-apod': expected 3 x {N} element normals, got {nrm.shape}")
            nrm = col(nrm)
        rxa = dict(kind=int(rxa["kind"]), p=tuple(float(v) for v in rxa.get("p", (0.0, 0.0))), normals=nrm)
    return DasProblem(fun=fun, prec=prec, flag=int(flag), VS=bool(opts["VS"]), DV=bool(opts["DV"]), Isz=Isz,
                      T=T, N=N, M=M, fsz=tuple(fsz), fs=fs, fmod=float(opts["fmod"]),
                      Pi=col(Pi), Pr=col(Pr), Pv=col(Pv4), Nv=col(Nv), cinv=col(cinv),
                      apod=flat, apod_real=apod_real, acstride=np.asarray(table, dtype=np.uint64),
                      S=len(apods), tpose=tpose, interp=interp, osize=osize, rx_apod=rxa)


# ------------------------------------------------------------------------------------------
# device side
# ------------------------------------------------------------------------------------------
def _torch():
    import torch
    return torch


def _colmajor(t):
    """Tensor whose memory is the column-major image of ``t`` (zero-copy if it already is).  Device tensors of up to three
    dimensions go through ``qdas_permute3`` (LDS-tiled, coalesced both ways); the rest through torch's strided copy."""
    if t.ndim <= 1:
        return t.contiguous()
    if t.is_cuda and 2 <= t.ndim <= 3 and t.is_contiguous() and t.element_size() in (2, 4, 8, 16) and t.numel():
        sh = tuple(t.shape) if t.ndim == 3 else (t.shape[0], 1, t.shape[1])
        if sh[1] <= 65535 and (sh[0] + 63) // 64 <= 65535:
            torch = _torch()
            out = torch.empty(tuple(reversed(t.shape)), dtype=t.dtype, device=t.device)
            with torch.cuda.device(t.device):
                _lib.check(_lib.lib().qdas_permute3(C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()), sh[0], sh[1], sh[2],
                                                    t.element_size(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            return out
    return t.permute(*reversed(range(t.ndim))).contiguous()


def _data_dtype(prec):
    torch = _torch()
    return {"double": torch.complex128, "single": torch.complex64, "halfT": torch.complex32}[prec]


def _cast_data(x, prec, device):
    """channel data -> complex(prec) on the device (reference kern/das_spec.m:243: dtypefun = complex(prec(x)))"""
    torch = _torch()
    if not _is_torch(x):
        x = torch.from_numpy(np.asarray(x))
    x = x.to(device)
    want = _data_dtype(prec)
    if x.dtype == want:
        return x
    if not x.is_complex():
        x = torch.complex(x.to(torch.float32 if prec != "double" else torch.float64),
                          torch.zeros((), device=device, dtype=torch.float32 if prec != "double" else torch.float64).expand_as(x))
    if prec == "halfT":
        return torch.view_as_complex(torch.view_as_real(x.to(torch.complex64)).to(torch.float16).contiguous())
    return x.to(want)


class DasPlan:
    """Reusable beamforming plan: the ``[k, PRE_ARGS, POST_ARGS]`` handle of the reference
    (``kern/das_spec.m:72-81,387-390``).  ``plan.feval(x)`` beamforms one ``T x N x M`` frame.

    A plan is NOT re-entrant: it owns scratch on the device (fallback-tile list, partial images of a split aperture), so frames go through
    it one after the other on ONE stream at a time -- as through the reference's kernel object.  ``das_spec`` hands the same cached plan to
    every call with an equal problem (``QDAS_PLAN_CACHE=0`` for a private plan per call); threads that beamform the same problem
    concurrently on different streams should hold their own ``DasPlan``."""

    def __init__(self, prob: DasProblem, device=None, kernel: int = _lib.KERNEL_AUTO,
                 i_begin: int = 0, i_count: int = 0, reciprocal: bool = True, jit: bool = False, mirror: bool = True, mirror_slab: bool = False,
                 fold: bool = True, approx_symmetry: bool = False, prefolded: bool = False):
        """``mirror_slab``: the slab ``[i_begin, i_begin + i_count)`` AND its mirror image in one plan (``QDAS_PLAN_MIRROR_SLAB``, ``include/qdas.h``):
        the output holds ``2 * i_count`` pixels, slab A then slab B; raises when the lateral-mirror mode is not available for the problem."""
        torch = _torch()
        self.lib = _lib.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("qups_amd: no HIP device visible -- the DAS path has no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.prob = prob
        self.i_begin = int(i_begin)
        self.i_count = int(i_count) if i_count else prob.I - int(i_begin)
        self.out_count = 2 * self.i_count if mirror_slab else self.i_count      # pixels per output plane
        dev = self.device
        up = lambda a: torch.from_numpy(a).to(dev) if a is not None else None
        self._bufs = [up(prob.Pi), up(prob.Pr), up(prob.Pv), up(prob.Nv), up(prob.cinv),
                      up(prob.apod.view(np.uint16) if (prob.apod is not None and prob.apod.dtype == np.float16) else prob.apod)]
        self._acs = (C.c_uint64 * len(prob.acstride))(*[int(v) for v in prob.acstride])
        d = _lib.Desc()
        d.sz = _lib.Sizes(prob.T, prob.N, prob.M, prob.Isz[0], prob.Isz[1], prob.Isz[2], prob.S, prob.flag,
                          int(prob.VS), int(prob.DV), _PREC[prob.prec])
        d.fs, d.fmod = prob.fs, prob.fmod
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
        d.Pi, d.Pr, d.Pv, d.Nv, d.cinv, d.apod = (ptr(self._bufs[0]), ptr(self._bufs[1]), ptr(self._bufs[2]),
                                                   ptr(self._bufs[3]), ptr(self._bufs[4]), ptr(self._bufs[5]))
        d.acstride = self._acs
        d.mem, d.apod_real, d.kernel = _lib.MEM_DEVICE, int(prob.apod_real), int(kernel)
        d.device = dev.index if dev.index is not None else torch.cuda.current_device()
        d.i_begin, d.i_count, d.y_ld = self.i_begin, self.i_count, 0
        d.plan_flags = ((0 if reciprocal else _lib.PLAN_NO_RECIPROCAL) | (_lib.PLAN_JIT if jit else 0) | (0 if mirror else _lib.PLAN_NO_MIRROR)
                        | (_lib.PLAN_MIRROR_SLAB if mirror_slab else 0) | (0 if fold else _lib.PLAN_NO_FOLD)
                        | (_lib.PLAN_APPROX_SYMMETRY if approx_symmetry else 0) | (_lib.PLAN_PREFOLDED if prefolded else 0))
        if prob.rx_apod is not None:                       # generated receive apodization (qdas.h QDAS_RXAPOD_*)
            d.rx_apod_kind = prob.rx_apod["kind"]
            d.rx_apod_p[0], d.rx_apod_p[1] = prob.rx_apod["p"]
            self._bufs.append(up(prob.rx_apod["normals"]))
            d.rx_normals = ptr(self._bufs[-1])
        self._desc = d
        self._h = C.c_void_p()
        self._lock = threading.RLock()          # execute / delays / close exclude each other: a cached plan evicted by another thread is never freed mid-call
        with torch.cuda.device(dev):
            _lib.check(self.lib.qdas_plan_create(C.byref(self._h), C.byref(d)))
            # why a plan that asked for a hiprtc build runs the stock kernel after all (no compiler, a build that would spill registers, ...): the
            # library leaves the reason in qdas_last_error() and carries on
            self._jit_note = (self.lib.qdas_last_error() or b"").decode(errors="replace") if jit else ""

    def jit_note(self) -> str:
        """``""``, or why ``jit=True`` did not give a plan-specialised kernel (``QDAS_PLAN_JIT: ... -- using the prebuilt kernel``)."""
        return self._jit_note if "QDAS_PLAN_JIT" in self._jit_note else ""

    # -- introspection
    @property
    def kernel(self) -> str:
        return _lib.KERNEL_NAMES.get(self.lib.qdas_plan_kernel(self._h), "?")

    def tile_shape(self) -> tuple:
        """(pixels of I1, columns) of one workgroup tile of the tiled kernel; (0, 0) for the generic kernel."""
        tz, tc = C.c_int(0), C.c_int(0)
        _lib.check(self.lib.qdas_plan_tile_shape(self._h, C.byref(tz), C.byref(tc), None, None))
        return int(tz.value), int(tc.value)

    def wave_shape(self) -> tuple:
        """(pixels of I1, columns) one wave covers inside a tile; (0, 0) for the generic kernel."""
        tz, tc, wz = C.c_int(0), C.c_int(0), C.c_int(0)
        _lib.check(self.lib.qdas_plan_tile_shape(self._h, C.byref(tz), C.byref(tc), C.byref(wz), None))
        return (int(wz.value), 64 // int(wz.value)) if wz.value else (0, 0)

    def aperture_split(self) -> int:
        """workgroups per tile (each sums a slice of the aperture; > 1 when the image / slab has too few tiles for the GPU)."""
        tz, tc, ks = C.c_int(0), C.c_int(0), C.c_int(0)
        _lib.check(self.lib.qdas_plan_tile_shape(self._h, C.byref(tz), C.byref(tc), None, C.byref(ks)))
        return int(ks.value)

    @property
    def reciprocal(self) -> bool:
        """True when the tiled kernel runs in reciprocal mode (FSA with transmit elements == receive elements, one t0)."""
        return bool(self.lib.qdas_plan_reciprocal(self._h))

    def symmetry_bound(self) -> tuple:
        """``(mirror, reciprocal)``: bounds [samples] of the delay error the plan's symmetry modes commit (``qdas_plan_symmetry_bound``): 0 exact symmetry,
        > 0 accepted within the tolerance of ``approx_symmetry=True`` (``QDAS_PLAN_APPROX_SYMMETRY``), -1 mode not in use"""
        a, b = C.c_double(), C.c_double()
        _lib.check(self.lib.qdas_plan_symmetry_bound(self._h, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    @property
    def folded(self) -> bool:
        """True when a reciprocal plan beamforms the reciprocity-folded frame (``qdas_plan_folded``: the two traces of every unordered
        transmit / receive pair are added once per frame, the fused kernel walks the upper triangle; ``fold=False`` / ``QDAS_PLAN_NO_FOLD``)."""
        return bool(self.lib.qdas_plan_folded(self._h))

    @property
    def mirror(self) -> bool:
        """True when the tiled kernel runs in lateral-mirror mode (scan, array and sequence mirror-symmetric about x = 0: a pixel and
        its mirror image share tap index and interpolation weights)."""
        return bool(self.lib.qdas_plan_mirror(self._h))

    def kernel_name(self) -> str:
        """name of the kernel the plan launches per frame, with ``[prebuilt]`` or ``[jit <hash>]`` (``qdas_plan_kernel_name``)"""
        buf = C.create_string_buffer(256)
        _lib.check(self.lib.qdas_plan_kernel_name(self._h, buf, 256))
        return buf.value.decode()

    def fallback_tiles(self) -> int:
        n = C.c_uint64()
        _lib.check(self.lib.qdas_plan_fallback_tiles(self._h, C.byref(n)))
        return int(n.value)

    def set_timing(self, on: bool = True):
        _lib.check(self.lib.qdas_plan_set_timing(self._h, int(on)))

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        _lib.check(self.lib.qdas_plan_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    # -- execution
    def _stream(self):
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def execute_colmajor(self, xc, F: int = 1):
        """``xc``: column-major channel data, i.e. a contiguous tensor shaped ``(F.., M, N, T)`` (or
        ``(F.., N, M, T)`` when transposed) of complex(prec).  Returns ``(F, oM, oN, i_count)``."""
        p = self.prob
        oN, oM = p.osize
        y = _torch().empty((F, oM, oN, self.out_count), dtype=_data_dtype(p.prec), device=self.device)
        return self.execute_into(xc, y, F)

    def execute_into(self, xc, y, F: int = 1):
        """:meth:`execute_colmajor` into a caller-owned output ``y`` (contiguous, ``F * oM * oN * i_count`` elements of complex(prec) on
        the plan's device): a frame stream reuses one image buffer instead of allocating per frame -- the reference's ``k.feval(yg, ...)``
        treats ``yg`` as in/out the same way (``kern/das_spec.m:349-351,372``).  Returns ``y``."""
        p = self.prob
        oN, oM = p.osize
        per = p.T * p.N * p.M
        want = _data_dtype(p.prec)
        if xc.numel() != F * per or not xc.is_contiguous():
            raise DasError("channel data size does not match the plan")
        if xc.dtype != want or xc.device != self.device:       # the library reads raw bytes: a wrong element size would run off the buffer
            raise DasError(f"channel data must be {want} on {self.device} for a '{p.prec}' plan, got {xc.dtype} on {xc.device}")
        if y.numel() != F * oM * oN * self.out_count or not y.is_contiguous() or y.dtype != want or y.device != self.device:
            raise DasError(f"output must be a contiguous {want} tensor of {F * oM * oN * self.out_count} elements on {self.device}")
        with self._lock:
            if self._h is None or not self._h.value:
                raise DasError("the plan has been closed")
            _lib.check(self.lib.qdas_plan_execute_frames(self._h, C.c_void_p(xc.data_ptr()), C.c_void_p(y.data_ptr()),
                                                         F, per, oM * oN * self.out_count, self._stream()))
        return y

    def prepare_frames(self, F: int):
        """Do now what the plan's first stream of ``F`` frames would do once (``qdas_plan_prepare_frames``: the second folded copy of a reciprocal plan, the
        frame-sharing kernel instantiations, a mirror plan's twin): afterwards :meth:`execute_into` with up to ``F`` frames only enqueues work."""
        with self._lock:
            if self._h is None or not self._h.value:
                raise DasError("the plan has been closed")
            _lib.check(self.lib.qdas_plan_prepare_frames(self._h, int(F)))
        return self

    def feval(self, x):
        """One frame ``x`` (``T x N x M``, MATLAB order) -> ``I x [1|N] x [1|M]`` like ``k.feval`` at
        reference ``kern/das_spec.m:372``."""
        xc = _colmajor(_cast_data(x, self.prob.prec, self.device))
        y = self.execute_colmajor(xc.reshape(1, *xc.shape[-3:]) if xc.ndim >= 3 else xc, 1)
        return y[0].permute(2, 1, 0)

    def delays(self):
        torch = _torch()
        p = self.prob
        rt = torch.float64 if p.prec == "double" else torch.float32
        tau = torch.empty((p.M, p.N, self.i_count), dtype=rt, device=self.device)
        with torch.cuda.device(self.device), self._lock:
            if self._h is None or not self._h.value:
                raise DasError("the plan has been closed")
            _lib.check(self.lib.qdas_plan_delays(self._h, C.c_void_p(tau.data_ptr()), self._stream()))
        return tau.permute(2, 1, 0)

    def close(self):
        """Destroy the native plan (its device allocations, events, staging buffers).  Work already queued on the plan's device is
        waited for first: the plan's tables must outlive the launches that read them."""
        lock = getattr(self, "_lock", None)
        if lock is None:                                # (constructor failed before the native plan existed)
            return
        with lock:                                      # (waits for a call of another thread that is inside the native plan)
            h = getattr(self, "_h", None)
            if h is not None and h.value:
                self._h = C.c_void_p()
                try:
                    _torch().cuda.synchronize(self.device)
                except Exception:                       # interpreter shutdown: torch may be half gone; hipFree synchronises anyway
                    pass
                self.lib.qdas_plan_destroy(h)

    @property
    def closed(self) -> bool:
        return not (getattr(self, "_h", None) is not None and self._h.value)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiDevicePlan:
    """One process, several GPUs: ``qdas_plan_create_sharded`` (``include/qdas.h``).  The image is split into ``len(devices)``
    contiguous slabs of the linear pixel index, one per entry of ``devices`` (an ordinal may repeat); the channel data handed to
    :meth:`feval` lives on ``devices[0]`` and is replicated by the library with peer copies; the result is the full image on
    ``devices[0]``.  The per-process layout (``torch.distributed``, one rank per GPU) is :mod:`qups_amd.dist`."""

    def __init__(self, prob: DasProblem, devices, kernel: int = _lib.KERNEL_AUTO, reciprocal: bool = True, jit: bool = False):
        torch = _torch()
        self.lib = _lib.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("qups_amd: no HIP device visible -- the DAS path has no CPU fallback")
        self.devices = [int(d) for d in devices]
        self.device = torch.device(f"cuda:{self.devices[0]}")
        self.prob = prob
        # (the same marshalling as DasPlan: device copies of the constant inputs on devices[0])
        dev = self.device
        up = lambda a: torch.from_numpy(a).to(dev) if a is not None else None
        self._bufs = [up(prob.Pi), up(prob.Pr), up(prob.Pv), up(prob.Nv), up(prob.cinv),
                      up(prob.apod.view(np.uint16) if (prob.apod is not None and prob.apod.dtype == np.float16) else prob.apod)]
        self._acs = (C.c_uint64 * len(prob.acstride))(*[int(v) for v in prob.acstride])
        d = _lib.Desc()
        d.sz = _lib.Sizes(prob.T, prob.N, prob.M, prob.Isz[0], prob.Isz[1], prob.Isz[2], prob.S, prob.flag,
                          int(prob.VS), int(prob.DV), _PREC[prob.prec])
        d.fs, d.fmod = prob.fs, prob.fmod
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
        d.Pi, d.Pr, d.Pv, d.Nv, d.cinv, d.apod = (ptr(b) for b in self._bufs[:6])
        d.acstride = self._acs
        d.mem, d.apod_real, d.kernel, d.device = _lib.MEM_DEVICE, int(prob.apod_real), int(kernel), self.devices[0]
        d.plan_flags = (0 if reciprocal else _lib.PLAN_NO_RECIPROCAL) | (_lib.PLAN_JIT if jit else 0)
        if prob.rx_apod is not None:
            d.rx_apod_kind = prob.rx_apod["kind"]
            d.rx_apod_p[0], d.rx_apod_p[1] = prob.rx_apod["p"]
            self._bufs.append(up(prob.rx_apod["normals"]))
            d.rx_normals = ptr(self._bufs[-1])
        self._desc = d
        self._h = C.c_void_p()
        devs = (C.c_int * len(self.devices))(*self.devices)
        with torch.cuda.device(dev):
            _lib.check(self.lib.qdas_plan_create_sharded(C.byref(self._h), C.byref(d), len(self.devices), devs))

    @property
    def mirror_slabs(self) -> bool:
        """True: every shard owns the slab :meth:`shards` reports AND its mirror image, the pixels ``[I - i_begin - i_count, I - i_begin)``"""
        return bool(self.lib.qdas_plan_sharded_mirror(self._h))

    def shards(self):
        """[(device, i_begin, i_count, kernel name)] per slab (with :attr:`mirror_slabs`: slab A of the shard; slab B is its mirror image)"""
        n = C.c_int()
        _lib.check(self.lib.qdas_plan_sharded_info(self._h, -1, C.byref(n), None, None, None))
        out = []
        for g in range(n.value):
            dv, b, c, k = C.c_int(), C.c_uint64(), C.c_uint64(), C.c_int()
            _lib.check(self.lib.qdas_plan_sharded_info(self._h, g, C.byref(dv), C.byref(b), C.byref(c), C.byref(k)))
            out.append((dv.value, int(b.value), int(c.value), _lib.KERNEL_NAMES.get(k.value, "-")))
        return out

    def execute_colmajor(self, xc):
        """one frame: ``xc`` column-major channel data ``(M, N, T)`` on ``devices[0]`` -> ``(oM, oN, I)`` there"""
        torch = _torch()
        p = self.prob
        oN, oM = p.osize
        if xc.numel() != p.T * p.N * p.M or not xc.is_contiguous() or xc.device != self.device or xc.dtype != _data_dtype(p.prec):
            raise DasError("channel data size / device / element type does not match the plan")
        y = torch.empty((oM, oN, p.I), dtype=_data_dtype(p.prec), device=self.device)
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(self.lib.qdas_plan_execute_sharded(self._h, C.c_void_p(xc.data_ptr()), C.c_void_p(y.data_ptr()), stream))
        return y

    def feval(self, x):
        """``x`` (``T x N x M``, MATLAB order) -> ``I x [1|N] x [1|M]`` like ``k.feval`` at reference ``kern/das_spec.m:372``"""
        xc = _colmajor(_cast_data(x, self.prob.prec, self.device))
        return self.execute_colmajor(xc.reshape(*xc.shape[-3:])).permute(2, 1, 0)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.qdas_plan_destroy_sharded(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------
# plan cache: the reference keeps its compiled kernel object `persistent` between calls (kern/wsinterpd2.m:51,181-190) and hands
# [k, PRE_ARGS, POST_ARGS] back for frame loops (kern/das_spec.m:72-81,387-390).  Here a call whose problem -- every argument except
# the channel data -- equals an earlier call's reuses that call's plan: no uploads, no probe launches, no table folding.
# ------------------------------------------------------------------------------------------
_PLAN_CACHE: "collections.OrderedDict[bytes, DasPlan]" = collections.OrderedDict()
_PLAN_CACHE_LOCK = threading.Lock()
_PLAN_CACHE_STATS = {"hits": 0, "misses": 0, "evictions": 0}


def _plan_cache_size() -> int:
    try:
        return max(0, int(os.environ.get("QDAS_PLAN_CACHE", "8")))
    except ValueError:
        return 8


def _memo_bytes() -> int:
    """host RAM the marshalling memos below may hold, EACH (``QDAS_HOST_MEMO_MB``, default 256; 0 switches them off): the column-major copies
    of geometry and apodization arrays a frame loop through ``das_spec`` would otherwise rebuild per call.  An array larger than the bound
    is simply not kept (BASELINE C5's 268 MB mask: set ``QDAS_HOST_MEMO_MB=1024``, or hold the plan -- the plan API needs none of this)."""
    try:
        return max(0, int(os.environ.get("QDAS_HOST_MEMO_MB", "256"))) << 20
    except ValueError:
        return 256 << 20


_COL_CACHE: "collections.OrderedDict[tuple, np.ndarray]" = collections.OrderedDict()
_COL_DIGEST: dict = {}
_MEMO_LOCK = threading.RLock()                            # the marshalling memos below are shared by every thread that calls das_spec


def _col_cached(A, rt):
    with _MEMO_LOCK:
        return _col_cached_locked(A, rt)


def _col_cached_locked(A, rt):
    """``A`` flattened in column-major order as ``rt`` -- memoised by CONTENT (a 64-bit xxh3 of the bytes, ~10 GB/s): a frame loop through
    ``das_spec`` / ``UltrasoundSystem.DAS`` hands over the same pixel grid every call, and the strided flatten of a 3 x 1024 x 1024 grid costs
    19 ms -- as much as beamforming the C3 frame.  (The reference keeps these arrays on the device between calls.)"""
    A = np.asarray(A)
    if A.size < 4096 or _memo_bytes() == 0:
        if A.size >= 4096:                                   # (memos switched off: drop what an earlier setting kept)
            _COL_CACHE.clear(); _FLAT_CACHE.clear(); _COL_DIGEST.clear()
        return np.ascontiguousarray(A.reshape(-1, order="F").astype(rt))
    Ac = np.ascontiguousarray(A)
    try:
        import xxhash
        dig = xxhash.xxh3_128(Ac.view(np.uint8).reshape(-1).data).digest()
    except ImportError:
        import hashlib
        dig = hashlib.blake2b(Ac.view(np.uint8).reshape(-1).data, digest_size=16).digest()
    key = (A.shape, A.dtype.str, np.dtype(rt).str, dig)
    hit = _COL_CACHE.get(key)
    if hit is not None:
        _COL_CACHE.move_to_end(key)
        return hit
    out = np.ascontiguousarray(Ac.reshape(-1, order="F").astype(rt))
    _COL_CACHE[key] = out
    _COL_DIGEST[id(out)] = (out, dig)                        # problem_key() takes the digest instead of hashing the flattened copy again
    while _COL_CACHE and (len(_COL_CACHE) > 16 or sum(v.nbytes for v in _COL_CACHE.values()) > _memo_bytes()):
        _, old = _COL_CACHE.popitem(last=False)
        _COL_DIGEST.pop(id(old), None)
    return out


def _immutable(a: np.ndarray) -> bool:
    """nobody can change the bytes behind ``a``: read-only, and so is every array it is a view of"""
    while isinstance(a, np.ndarray):
        if a.flags.writeable:
            return False
        a = a.base
    return a is None or isinstance(a, (bytes, memoryview)) and getattr(a, "readonly", True)


_FLAT_CACHE: "collections.OrderedDict[tuple, tuple]" = collections.OrderedDict()


def _flat_cached(a: np.ndarray, adt) -> np.ndarray:
    with _MEMO_LOCK:
        return _flat_cached_locked(a, adt)


def _flat_cached_locked(a: np.ndarray, adt) -> np.ndarray:
    """``_flat_colmajor(a, adt)`` memoised for large apodization arrays: by content, or -- opt-in, ``QDAS_HOST_MEMO_BY_IDENTITY=1`` -- by
    IDENTITY when the array is immutable (``a.setflags(write=False)``: the generators of ``qups_amd.apodization`` return such arrays; by content = xxh3 of the bytes: 27 ms for BASELINE C5's 268 MB mask, where
    the transposing cast takes 60-130 ms).  A frame loop through ``das_spec`` / ``UltrasoundSystem.DAS`` hands over the same mask every call; the
    plan API (``DasPlan`` / ``return_plan=True``) skips all of this."""
    if a.size < (1 << 16) or _memo_bytes() == 0:
        return _flat_colmajor(a, adt)
    ident = None
    # (opt-in: a caller who flips the write flag, edits and flips it back would be served the OLD weights -- nothing but the content can tell)
    if os.environ.get("QDAS_HOST_MEMO_BY_IDENTITY", "0") not in ("", "0") and _immutable(a):
        # (the cache entry keeps a reference to the array: its memory cannot be freed and handed to another array while the entry lives)
        ident = ("buffer", a.__array_interface__["data"][0], a.shape, a.strides, a.dtype.str, np.dtype(adt).str)
        hit = _FLAT_CACHE.get(ident)
        if hit is not None:
            _FLAT_CACHE.move_to_end(ident)
            return hit[1]
    ac = np.ascontiguousarray(a)
    h = _hasher()
    h.update(ac.view(np.uint8).reshape(-1).data)
    dig = h.digest()
    key = ("content", a.shape, a.dtype.str, np.dtype(adt).str, dig)
    hit = _FLAT_CACHE.get(key)
    if hit is None:
        out = _flat_colmajor(a, adt)
        _FLAT_CACHE[key] = (None, out)
        _COL_DIGEST[id(out)] = (out, dig)
    else:
        _FLAT_CACHE.move_to_end(key)
        out = hit[1]
    if ident is not None:
        _FLAT_CACHE[ident] = (a, out)
    while _FLAT_CACHE and (len(_FLAT_CACHE) > 12 or sum(v[1].nbytes for v in {id(v[1]): v for v in _FLAT_CACHE.values()}.values()) > _memo_bytes()):
        _, old = _FLAT_CACHE.popitem(last=False)
        if not any(v[1] is old[1] for v in _FLAT_CACHE.values()):
            _COL_DIGEST.pop(id(old[1]), None)
    return out


def _hasher():
    try:                                    # xxh3: ~10 GB/s (BASELINE C5's 268 MB mask in 25 ms); blake2b as the portable stand-in
        import xxhash
        return xxhash.xxh3_128()
    except ImportError:
        import hashlib
        return hashlib.blake2b(digest_size=16)


def _qdas_env():
    """the ``QDAS_*`` environment as a sorted tuple (from the raw byte table where there is one: no decode of the other ~100 variables per call)"""
    raw = getattr(os.environ, "_data", None)
    if isinstance(raw, dict) and raw and isinstance(next(iter(raw)), bytes):
        return tuple(sorted((k, v) for k, v in raw.items() if k.startswith(b"QDAS_") and k != b"QDAS_PLAN_CACHE" and not k.startswith(b"QDAS_HOST_MEMO")))
    return tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("QDAS_") and k != "QDAS_PLAN_CACHE" and not k.startswith("QDAS_HOST_MEMO")))


def problem_key(prob: DasProblem, *extra) -> bytes:
    """Digest of everything a plan is built from: sizes, flags, geometry, sound speed, apodization (contents, not identities), plus
    ``extra`` (device, kernel choice, plan flags) and the ``QDAS_*`` environment, which steers plan construction."""
    h = _hasher()
    head = (prob.fun, prob.prec, prob.flag, prob.VS, prob.DV, prob.Isz, prob.T, prob.N, prob.M, prob.fs, prob.fmod, prob.apod_real,
            prob.S, prob.tpose, prob.interp, prob.osize, extra,
            _qdas_env())
    h.update(repr(head).encode())
    arrays = [prob.Pi, prob.Pr, prob.Pv, prob.Nv, prob.cinv, prob.apod, prob.acstride]
    if prob.rx_apod is not None:
        h.update(repr((prob.rx_apod["kind"], prob.rx_apod["p"])).encode())
        arrays.append(prob.rx_apod["normals"])
    for a in arrays:
        if a is None:
            h.update(b"\0none")
        else:
            with _MEMO_LOCK:
                known = _COL_DIGEST.get(id(a))
            if known is not None and known[0] is a:              # a memoised flatten: its content digest (of the source array) stands for it
                h.update(repr((a.dtype.str, a.shape)).encode())
                h.update(b"digest" + known[1])
                continue
            a = np.ascontiguousarray(a)
            h.update(repr((a.dtype.str, a.shape)).encode())
            h.update(a.view(np.uint8).reshape(-1).data if a.size else b"")
    return h.digest()


def clear_plan_cache():
    """Destroy every cached plan (their device memory is released)."""
    with _PLAN_CACHE_LOCK:
        plans = list(_PLAN_CACHE.values())
        _PLAN_CACHE.clear()
    for p in plans:
        p.close()


def plan_cache_info() -> dict:
    with _PLAN_CACHE_LOCK:
        return dict(_PLAN_CACHE_STATS, size=len(_PLAN_CACHE), capacity=_plan_cache_size())


def _cached_plan(prob: DasProblem, device, kernel, jit):
    """(plan, owned): the cached plan of an equal problem, else a new one.  ``owned``: the plan is NOT in the cache (cache disabled)
    and the caller must close it unless it hands it out."""
    cap = _plan_cache_size()
    if cap == 0:
        return DasPlan(prob, device=device, kernel=kernel, jit=jit), True
    torch = _torch()
    dev = device if device is not None else (f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else None)
    key = problem_key(prob, str(dev), int(kernel), bool(jit))
    with _PLAN_CACHE_LOCK:
        plan = _PLAN_CACHE.get(key)
        if plan is not None and not plan.closed:
            _PLAN_CACHE.move_to_end(key)
            _PLAN_CACHE_STATS["hits"] += 1
            return plan, False
        _PLAN_CACHE_STATS["misses"] += 1
    plan = DasPlan(prob, device=device, kernel=kernel, jit=jit)
    evicted = []
    with _PLAN_CACHE_LOCK:
        _PLAN_CACHE[key] = plan
        _PLAN_CACHE.move_to_end(key)
        while len(_PLAN_CACHE) > cap:
            evicted.append(_PLAN_CACHE.popitem(last=False)[1])
            _PLAN_CACHE_STATS["evictions"] += 1
    for p in evicted:                       # (close() waits for the plan's device: a frame still in flight keeps its tables)
        p.close()
    return plan, False


def das_spec(fun, Pi, Pr, Pv, Nv, x, t0, fs=None, c=None, *varargin, return_plan=False,
             kernel: int = _lib.KERNEL_AUTO, jit: bool | None = None):
    """``y = das_spec(fun, Pi, Pr, Pv, Nv, x, t0, fs, c, ...)`` -- see reference ``kern/das_spec.m:1-87``.

    ``jit``: build the fused kernel for this call's sizes through hiprtc (``QDAS_PLAN_JIT``; the reference compiles its kernel per
    call with the sizes as constants, ``src/UltrasoundSystem.m:5626-5748``): first use of a shape costs a few seconds, later ones
    load from ``~/.cache/qdas``; 5-8 % faster than the prebuilt kernels.  Default: the environment variable ``QDAS_JIT`` (off).

    ``fun`` in ``{'DAS','SYN','MUL','BF','delays'}``; options (strings, as in the reference):
    ``'plane-waves' | 'virtual-source' | 'diverging-waves' | 'focused-waves'``,
    ``'input-precision', {'double','single','halfT'}``, ``'device', id``, ``'interp', method``,
    ``'apod', A`` (repeatable), ``'modulation', fmod``, ``'transpose', tf``.

    Returns ``y`` (``I1 x I2 x I3 x [1|N] x [1|M] x F...``, torch tensor on the device); with
    ``return_plan=True`` also the reusable :class:`DasPlan` (the reference's 2nd-4th outputs).
    ``'device', 0`` asks the reference for its native-MATLAB CPU branch; this package IS the
    device path and raises instead of silently computing on the host.
    """
    opts = parse_options(x, varargin)
    if opts["device"] == 0:
        raise NotImplementedError("das_spec(..., 'device', 0): qups_amd implements the device path only "
                                  "(no CPU fallback by design)")
    torch = _torch()
    xshape = tuple(x.shape) if fun != "delays" else ()
    prob = build_problem(fun, Pi, Pr, Pv, Nv, xshape, t0, fs, c, opts)
    # MATLAB device ids are 1-based; any negative id means "the current device" (the reference passes -1, kern/das_spec.m:131)
    device = None if (opts["device"] is None or opts["device"] < 0) else f"cuda:{opts['device'] - 1}"
    if jit is None:
        jit = os.environ.get("QDAS_JIT", "0") not in ("", "0")
    # jit: hiprtc build for these sizes (qdas.h QDAS_PLAN_JIT).  The plan comes from the cache when an equal problem has been seen
    # (QDAS_PLAN_CACHE = number of plans kept, default 8; 0: a fresh plan per call, destroyed before returning unless handed out)
    plan, owned = _cached_plan(prob, device, kernel, jit)
    try:
        Isz = prob.Isz
        rev = lambda t: t.permute(*reversed(range(t.ndim)))
        if fun == "delays":
            tau = rev(plan.delays()).contiguous()           # (M, N, I) column-major parent
            y = rev(tau.reshape(prob.M, prob.N, Isz[2], Isz[1], Isz[0]))
        else:
            xd = _cast_data(x, prob.prec, plan.device)
            while xd.ndim < 3:
                xd = xd.unsqueeze(-1)
            F = int(np.prod(prob.fsz)) if prob.fsz else 1
            xc = _colmajor(xd)                                  # (F.., M, N, T): MATLAB memory order
            yc = plan.execute_colmajor(xc, F)                   # (F, oM, oN, I)
            oN, oM = prob.osize
            y = rev(yc.reshape(tuple(reversed(prob.fsz)) + (oM, oN, Isz[2], Isz[1], Isz[0])))
    except BaseException:
        if owned:
            plan.close()
        raise
    if return_plan:
        # A plan that is handed out belongs to the caller from here on: it leaves the cache, so that no later call shares its scratch
        # (fallback list, partial images) on another stream and no LRU eviction closes it under the caller's frame loop.
        if not owned:
            with _PLAN_CACHE_LOCK:
                for k, v in list(_PLAN_CACHE.items()):
                    if v is plan:
                        del _PLAN_CACHE[k]
        return y, plan
    if owned:
        plan.close()
    return y                                            # I1 x I2 x I3 x [1|N] x [1|M] x F...
