"""ctypes binding of ``libqdas.so`` -- the only way the Python host reaches the HIP kernels.

Mirrors ``include/qdas.h`` one to one.  There is deliberately NO fallback: if the shared
library is missing or does not export a symbol, importing this module's :func:`lib`
raises; if no HIP device is present, plan creation fails with the HIP error text.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QDAS_LIB") or os.path.join(_HERE, "libqdas.so")   # QDAS_LIB: A/B builds for profiling

# ---- constants (include/qdas.h)
QDAS_F64, QDAS_F32, QDAS_F16 = 0, 1, 2
INTERP_FLAGS = {"nearest": 0, "linear": 1, "cubic": 2, "lanczos3": 3, "cubic_dev": 5}
FLAG_KEEP_RX, FLAG_KEEP_TX, FLAG_TPOSE = 8, 16, 32
MEM_HOST, MEM_DEVICE = 0, 1
KERNEL_AUTO, KERNEL_GENERIC, KERNEL_TILED = 0, 1, 2
KERNEL_NAMES = {KERNEL_GENERIC: "generic", KERNEL_TILED: "tiled"}
MAX_APOD = 6
QDAS_PRE_F32, QDAS_PRE_I16 = 0, 1
QDAS_CONV_FULL, QDAS_CONV_SAME, QDAS_CONV_VALID, QDAS_CONV_CAUSAL = 0, 1, 2, 3
QDAS_CONV_X_ONE_COLUMN, QDAS_CONV_X_ONE_SLICE, QDAS_CONV_Y_ONE_COLUMN, QDAS_CONV_Y_ONE_SLICE = 1, 2, 4, 8
PLAN_NO_RECIPROCAL, PLAN_JIT, PLAN_COPY_INPUTS, PLAN_NO_MIRROR, PLAN_MIRROR_SLAB, PLAN_NO_FOLD, PLAN_APPROX_SYMMETRY, PLAN_PREFOLDED = 1, 2, 4, 8, 16, 32, 64, 128
RXAPOD_NONE, RXAPOD_ACCEPTANCE, RXAPOD_COSINE, RXAPOD_FNUMBER_PLANAR, RXAPOD_FNUMBER_ORIENTED = 0, 1, 2, 3, 4

# every symbol include/qdas.h declares (tests check the library exports all of them)
SYMBOLS = (
    "qdas_plan_create", "qdas_plan_execute", "qdas_plan_execute_frames", "qdas_plan_prepare_frames", "qdas_plan_delays",
    "qdas_plan_destroy", "qdas_plan_kernel", "qdas_plan_fallback_tiles", "qdas_plan_tile_shape", "qdas_plan_reciprocal", "qdas_plan_folded", "qdas_fold", "qdas_plan_symmetry_bound", "qdas_plan_mirror", "qdas_plan_kernel_name", "qdas_plan_set_timing",
    "qdas_plan_last_kernel_ms", "qdas_plan_create_sharded", "qdas_plan_execute_sharded", "qdas_plan_sharded_info", "qdas_plan_sharded_mirror",
    "qdas_plan_destroy_sharded", "qdas_DAS", "qdas_DASf", "qdas_DASh", "qdas_delays", "qdas_delaysf",
    "qdas_das_lut", "qdas_das_lut_last_kernel", "qdas_wsinterpd", "qdas_shift_sum", "qdas_greens", "qdas_convd", "qdas_convd_len", "qdas_permute3", "qdas_pre_plan_create", "qdas_pre_execute", "qdas_pre_plan_destroy", "qdas_pre_plan_one_pass", "qdas_last_error", "qdas_version", "qdas_device_malloc", "qdas_device_free", "qdas_device_trim", "qdas_device_copy", "qdas_iir", "qdas_device_info", "qdas_kernel_variant_build", "qdas_kernel_variant_prebuilt",
)


class Sizes(C.Structure):
    _fields_ = [("T", C.c_uint64), ("N", C.c_uint64), ("M", C.c_uint64),
                ("I1", C.c_uint64), ("I2", C.c_uint64), ("I3", C.c_uint64), ("S", C.c_uint64),
                ("flag", C.c_int32), ("VS", C.c_int32), ("DV", C.c_int32), ("dtype", C.c_int32)]


class Desc(C.Structure):
    _fields_ = [("sz", Sizes), ("fs", C.c_double), ("fmod", C.c_double),
                ("Pi", C.c_void_p), ("Pr", C.c_void_p), ("Pv", C.c_void_p), ("Nv", C.c_void_p),
                ("apod", C.c_void_p), ("cinv", C.c_void_p), ("acstride", C.POINTER(C.c_uint64)),
                ("mem", C.c_int32), ("apod_real", C.c_int32), ("kernel", C.c_int32), ("device", C.c_int32),
                ("i_begin", C.c_uint64), ("i_count", C.c_uint64), ("y_ld", C.c_uint64),
                ("rx_apod_kind", C.c_int32), ("plan_flags", C.c_int32), ("rx_apod_p", C.c_double * 2),
                ("rx_normals", C.c_void_p)]


class LutDesc(C.Structure):
    _fields_ = [("T", C.c_uint64), ("N", C.c_uint64), ("M", C.c_uint64), ("I", C.c_uint64),
                ("flag", C.c_int32), ("dtype", C.c_int32), ("omega", C.c_double),
                ("tau_rx", C.c_void_p), ("tau_tx", C.c_void_p), ("w", C.c_void_p),
                ("wstride", C.c_uint64 * 3), ("w_real", C.c_int32), ("reserved", C.c_int32), ("I1", C.c_uint64)]


class WsDesc(C.Structure):
    _fields_ = [("T", C.c_uint64), ("x_tstride", C.c_uint64), ("ndim", C.c_int32), ("flag", C.c_int32), ("dtype", C.c_int32),
                ("w_real", C.c_int32), ("size", C.c_uint64 * 8), ("tstride", C.c_int64 * 8), ("xstride", C.c_int64 * 8),
                ("wstride", C.c_int64 * 8), ("sum", C.c_uint8 * 8), ("omega", C.c_double), ("extrap", C.c_double),
                ("t", C.c_void_p), ("w", C.c_void_p), ("x", C.c_void_p), ("ystride", C.c_int64 * 8), ("lane_dim", C.c_int32), ("reserved", C.c_int32)]


class FoldDesc(C.Structure):
    _fields_ = [("T", C.c_uint64), ("N", C.c_uint64), ("strN", C.c_uint64), ("strM", C.c_uint64), ("dtype", C.c_int32), ("device", C.c_int32), ("wtab", C.c_void_p)]


class IirDesc(C.Structure):
    _fields_ = [("T", C.c_uint64), ("K", C.c_uint64), ("nsec", C.c_int32), ("dtype", C.c_int32), ("cplx", C.c_int32), ("device", C.c_int32),
                ("gain", C.c_double), ("sos", C.POINTER(C.c_double))]


class GreensDesc(C.Structure):
    _fields_ = [("S", C.c_uint64), ("T", C.c_uint64), ("N", C.c_uint64), ("M", C.c_uint64), ("I", C.c_uint64),
                ("En", C.c_int32), ("Em", C.c_int32), ("interp", C.c_int32), ("dtype", C.c_int32),
                ("s0", C.c_double), ("t0", C.c_double), ("fs", C.c_double), ("fsr", C.c_double), ("cinv", C.c_double),
                ("R0", C.c_double), ("Ps", C.c_void_p), ("a", C.c_void_p), ("Pr", C.c_void_p), ("Pv", C.c_void_p),
                ("x", C.c_void_p), ("device", C.c_int32), ("reserved", C.c_int32)]


class PreDesc(C.Structure):
    _fields_ = [("T", C.c_uint64), ("K", C.c_uint64), ("Nfft", C.c_uint64), ("in_type", C.c_int32), ("device", C.c_int32),
                ("fs", C.c_double), ("t0", C.c_double), ("fdown", C.c_double)]


class ShiftDesc(C.Structure):
    _fields_ = [("T", C.c_uint64), ("To", C.c_uint64), ("N", C.c_uint64), ("M", C.c_uint64), ("Mo", C.c_uint64), ("F", C.c_uint64),
                ("flag", C.c_int32), ("dtype", C.c_int32), ("cplx", C.c_int32), ("w_real", C.c_int32), ("device", C.c_int32), ("tpad", C.c_int32),
                ("shift", C.c_void_p), ("w", C.c_void_p)]


class ConvdDesc(C.Structure):
    _fields_ = [("C", C.c_uint64), ("M", C.c_uint64), ("N", C.c_uint64), ("S", C.c_uint64), ("dtype", C.c_int32), ("cplx", C.c_int32),
                ("shape", C.c_int32), ("bcast", C.c_int32), ("device", C.c_int32), ("y_real", C.c_int32)]


class QdasError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libqdas error {code}: {msg}")
        self.code = code
        self.message = msg


_LIB = None


def lib():
    """Load ``libqdas.so`` (built in-tree by ``__graft_entry__.build()`` / ``make -C qups_amd/csrc``)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C qups_amd/csrc`). There is no CPU fallback in this package.")
    try:  # share torch's HIP runtime (same SONAME libamdhip64.so.7) when torch is in the process
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - the library also works without torch (MEX / C callers)
        pass
    L = C.CDLL(LIB_PATH)
    missing = [s for s in SYMBOLS if not hasattr(L, s)]
    if missing:
        raise ImportError(f"{LIB_PATH} does not export {missing}")
    L.qdas_last_error.restype = C.c_char_p
    L.qdas_plan_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(Desc)]
    L.qdas_plan_execute.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.qdas_plan_execute_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
    L.qdas_plan_prepare_frames.argtypes = [C.c_void_p, C.c_uint64]
    L.qdas_plan_delays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.qdas_plan_destroy.argtypes = [C.c_void_p]
    L.qdas_plan_destroy.restype = None
    L.qdas_plan_kernel.argtypes = [C.c_void_p]
    L.qdas_plan_fallback_tiles.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.qdas_plan_tile_shape.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.qdas_plan_reciprocal.argtypes = [C.c_void_p]
    L.qdas_plan_mirror.argtypes = [C.c_void_p]
    L.qdas_plan_folded.argtypes = [C.c_void_p]
    L.qdas_fold.argtypes = [C.POINTER(FoldDesc), C.c_void_p, C.c_void_p, C.c_void_p]
    L.qdas_plan_symmetry_bound.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.qdas_plan_kernel_name.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.qdas_plan_set_timing.argtypes = [C.c_void_p, C.c_int]
    L.qdas_plan_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.qdas_plan_create_sharded.argtypes = [C.POINTER(C.c_void_p), C.POINTER(Desc), C.c_int, C.POINTER(C.c_int)]
    L.qdas_plan_execute_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.qdas_plan_sharded_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    L.qdas_plan_sharded_mirror.argtypes = [C.c_void_p]
    L.qdas_plan_destroy_sharded.argtypes = [C.c_void_p]
    L.qdas_plan_destroy_sharded.restype = None
    L.qdas_das_lut.argtypes = [C.POINTER(LutDesc), C.c_void_p, C.c_void_p, C.c_void_p]
    L.qdas_wsinterpd.argtypes = [C.POINTER(WsDesc), C.c_void_p, C.c_void_p]
    L.qdas_greens.argtypes = [C.POINTER(GreensDesc), C.c_void_p, C.c_void_p]
    L.qdas_convd.argtypes = [C.POINTER(ConvdDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.qdas_permute3.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
    L.qdas_convd_len.argtypes = [C.c_uint64, C.c_uint64, C.c_int]
    L.qdas_convd_len.restype = C.c_uint64
    L.qdas_shift_sum.argtypes = [C.POINTER(ShiftDesc), C.c_void_p, C.c_void_p, C.c_void_p]
    L.qdas_pre_plan_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(PreDesc)]
    L.qdas_pre_execute.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.qdas_pre_plan_destroy.argtypes = [C.c_void_p]
    L.qdas_pre_plan_destroy.restype = None
    L.qdas_pre_plan_one_pass.argtypes = [C.c_void_p]
    L.qdas_kernel_variant_build.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    L.qdas_kernel_variant_prebuilt.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    L.qdas_device_malloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_int]
    L.qdas_device_free.argtypes = [C.c_void_p, C.c_int]
    L.qdas_device_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
    L.qdas_device_trim.argtypes = []
    L.qdas_device_info.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_uint64)]
    vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
    for name in ("qdas_DAS", "qdas_DASf", "qdas_DASh"):
        getattr(L, name).argtypes = [C.POINTER(Sizes), vp, vp, vp, vp, vp, vp, vp, u64p, vp, vp, vp]
    L.qdas_delays.argtypes = [C.POINTER(Sizes), vp, vp, vp, vp, vp, C.c_double, vp]
    L.qdas_delaysf.argtypes = [C.POINTER(Sizes), vp, vp, vp, vp, vp, C.c_float, vp]
    _LIB = L
    return L


def check(rc: int):
    if rc:
        raise QdasError(rc, (lib().qdas_last_error() or b"").decode(errors="replace"))


def device_info(device: int = -1) -> dict:
    name = C.create_string_buffer(256)
    cu, clk, mem = C.c_int(), C.c_int(), C.c_uint64()
    check(lib().qdas_device_info(device, name, 256, C.byref(cu), C.byref(clk), C.byref(mem)))
    return {"name": name.value.decode(), "cu_count": cu.value, "clock_khz": clk.value, "hbm_bytes": mem.value}
