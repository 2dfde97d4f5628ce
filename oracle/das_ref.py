"""ctypes front-end for the C oracle ``oracle/libdas_ref.so`` (TEST INFRASTRUCTURE).

Marshals MATLAB-ordered numpy arrays into the column-major buffers + stride
table of the reference kernel ABI (``kern/das_spec.m:257-260,344-345,361``) and
calls ``das_ref_f32`` / ``das_ref_f64``.  Independent of ``qups_amd`` on purpose:
it is a second statement of the same marshalling, so tests that compare the
product with this oracle also cross-check the stride tables.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
LAST_SECONDS = 0.0   # wall time of the last C call (bench.py cpu_baseline)


class Sizes(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("T", "N", "M", "I", "I1", "I2", "I3", "S")] + [
        ("flag", C.c_int32), ("VS", C.c_int32), ("DV", C.c_int32)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libdas_ref.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


def _host_tag() -> str:
    """identifies the host's instruction set: the tuned build is -march=native and must never run on another CPU model"""
    import hashlib
    try:
        txt = open("/proc/cpuinfo").read()
        flags = next((l for l in txt.splitlines() if l.startswith("flags")), "")
        model = next((l for l in txt.splitlines() if l.startswith("model name")), "")
    except OSError:
        flags = model = ""
    return hashlib.sha1((model + flags).encode()).hexdigest()[:12]


def build_tuned() -> str:
    """``das_ref_tuned.c`` compiled FOR THIS HOST (gcc -O3 -march=native -fopenmp, no -ffast-math) at first use; cached per CPU
    model under ``oracle/_tuned/`` (git- and gpurun-ignored: a build made elsewhere may use instructions this host lacks)."""
    d = os.path.join(_HERE, "_tuned")
    so = os.path.join(d, f"libdas_ref_tuned.{_host_tag()}.so")
    src = os.path.join(_HERE, "das_ref_tuned.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(d, exist_ok=True)
        tmp = so + f".{os.getpid()}.tmp"
        subprocess.check_call([os.environ.get("CC", "gcc"), "-O3", "-march=native", "-fopenmp", "-fPIC", "-shared", "-Wall",
                               "-Wno-unknown-pragmas", "-o", tmp, src, "-lm"])
        os.replace(tmp, so)
    return so


_LIB_TUNED = None


def lib(tuned: bool = False):
    global _LIB, _LIB_TUNED
    if tuned:
        if _LIB_TUNED is None:
            _LIB_TUNED = C.CDLL(build_tuned())
            _LIB_TUNED.das_ref_tuned_f32.restype = C.c_int
            _LIB_TUNED.das_ref_max_threads.restype = C.c_int
        return _LIB_TUNED
    if _LIB is None:
        _LIB = C.CDLL(build())
        for name in ("das_ref_f32", "das_ref_f64"):
            getattr(_LIB, name).restype = C.c_int
        _LIB.das_ref_max_threads.restype = C.c_int
    return _LIB


def _fcol(a, dt):
    """column-major flat copy"""
    return np.ascontiguousarray(np.asarray(a, dtype=dt).reshape(-1, order="F"))


def _strides(shape5):
    """element strides, 0 on singleton dims (kern/das_spec.m:259-260)"""
    st, acc = [], 1
    for s in shape5:
        st.append(0 if s == 1 else acc)
        acc *= s
    return st


def _pad5(a):
    a = np.asarray(a)
    while a.ndim < 5:
        a = a[..., None]
    return a


def das_spec(fun, Pi, Pr, Pv, Nv, x, t0, fs, c=1540.0, *, VS=True, DV=False, interp="linear",
             apod=(), fmod=0.0, tpose=False, prec="single", nthreads=0, timing=False, tuned=False):
    """Same signature/result as :func:`oracle.das_oracle.das_spec` (single frame,
    device ``fmod`` semantics), computed by the C oracle in ``prec``."""
    rt = np.float32 if prec == "single" else np.float64
    ct = np.complex64 if prec == "single" else np.complex128
    flagn = {"nearest": 0, "linear": 1, "cubic": 2, "lanczos3": 3, "cubic_dev": 5}[interp]
    keep_rx, keep_tx = fun in ("SYN", "BF"), fun in ("MUL", "BF")
    Pi = np.asarray(Pi, dtype=rt)
    while Pi.ndim < 4:
        Pi = Pi[..., None]
    Isz = Pi.shape[1:4]
    I = int(np.prod(Isz))
    x = np.asarray(x)
    T = x.shape[0]
    N, M = (x.shape[2], x.shape[1]) if tpose else (x.shape[1], x.shape[2])
    Pr = np.broadcast_to(np.asarray(Pr, rt).reshape(3, -1), (3, N))
    Pv = np.broadcast_to(np.asarray(Pv, rt).reshape(3, -1), (3, M))
    Nv = np.broadcast_to(np.asarray(Nv, rt).reshape(3, -1), (3, M))
    Pv4 = np.concatenate([Pv, np.broadcast_to(np.asarray(t0, rt).reshape(1, -1), (1, M))], 0)
    cinv = _pad5((1.0 / np.asarray(c, dtype=np.float64)).astype(rt))
    aps = [_pad5(np.asarray(a).astype(ct)) for a in apod]
    table = _strides(cinv.shape) + [0]
    base = 0
    for a in aps:
        table += _strides(a.shape) + [base]
        base += a.size
    acs = np.asarray(table, dtype=np.uint64)
    apbuf = (np.concatenate([_fcol(a, ct) for a in aps]) if aps else np.zeros(1, ct)).view(rt)
    sz = Sizes(T, N, M, I, Isz[0], Isz[1], Isz[2], len(aps),
               flagn + 8 * keep_rx + 16 * keep_tx + 32 * bool(tpose), int(VS), int(DV))
    oN, oM = (N if keep_rx else 1), (M if keep_tx else 1)
    y = np.zeros(I * oN * oM, dtype=ct)
    bufs = [_fcol(Pi, rt), _fcol(Pr, rt), _fcol(Pv4, rt), _fcol(Nv, rt), apbuf,
            _fcol(cinv, rt), acs, _fcol(x.reshape(x.shape[:3]), ct).view(rt),
            np.asarray([fs, fmod], dtype=rt)]
    fn = lib().das_ref_f32 if prec == "single" else lib().das_ref_f64
    import time
    global LAST_SECONDS
    rc = 3
    if tuned:                                       # host-tuned build of the same loop nest (das_ref_tuned.c); rc 3 = outside its subset
        if prec != "single":
            raise ValueError("the tuned baseline is fp32 only")
        t_start = time.perf_counter()
        rc = lib(True).das_ref_tuned_f32(C.byref(sz), y.ctypes.data_as(C.c_void_p), *[b.ctypes.data_as(C.c_void_p) for b in bufs],
                                         C.c_int(nthreads))
        LAST_SECONDS = time.perf_counter() - t_start
        if rc == 3:
            raise ValueError("das_ref_tuned covers 'DAS', fp32, scalar sound speed, no apodization, fmod = 0 only")
    else:
        t_start = time.perf_counter()
        rc = fn(C.byref(sz), y.ctypes.data_as(C.c_void_p), *[b.ctypes.data_as(C.c_void_p) for b in bufs],
                C.c_int(nthreads))
        LAST_SECONDS = time.perf_counter() - t_start
    if rc:
        raise RuntimeError(f"das_ref failed rc={rc}")
    if keep_rx and keep_tx and tpose:   # 'BF' keeps the data's aperture order: I x M x N (src/bf.cu:100,135)
        oN, oM = oM, oN
    return y.reshape(tuple(Isz) + (oN, oM), order="F")
