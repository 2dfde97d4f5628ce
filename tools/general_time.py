#!/usr/bin/env python3
"""The general (non-fused) kernels that ChannelData.sample / rectifyt0 / focusTx / bfDAS fall onto, timed on shapes of the BASELINE
configurations with a bytes-based roofline each:  tools/general_time.py  ->  one line per case:
   ms, algorithmic GB moved (inputs read once + outputs written once), GB/s, fraction of the 8 TB/s HBM roof.
Times are torch.cuda.Event pairs on torch's current stream -- the stream every call here launches on -- around four calls issued back to back."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qups_amd.interpd import wsinterpd, das_lut

dev = torch.device("cuda:0")
HBM = 8000.0


def timed(fn, reps=5, inner=4):
    """median over `reps` of the time per call of `inner` calls issued back to back (a stream of calls: the device works on call k while the host prepares
    call k + 1; one call between two events on an idle device also counts the ~0.07 ms the host needs to get the first launch out)"""
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(reps):
        e0.record()
        for _ in range(inner):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / inner)
    return float(np.median(ms))


def line(name, ms, nbytes, note=""):
    gbs = nbytes / ms / 1e6
    print(f"{name:58s} {ms:9.3f} ms  {nbytes / 1e9:7.3f} GB  {gbs:8.1f} GB/s  {gbs / HBM:6.3f} of HBM roof  {note}")


g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.view_as_complex(torch.randn(tuple(s) + (2,), generator=g, device=dev, dtype=torch.float32))
# the same in MATLAB's (column-major) memory order -- time fastest --, which is how the reference hands its records to these kernels; a torch-order
# (last dimension fastest) record summed over its fastest dimension takes the lane-sum kernel (csrc/wsinterpd.hip; QDAS_WS_NO_LANESUM=1: transposed first, round 5)
rn_cm = lambda *s: rn(*reversed(s)).permute(*reversed(range(len(s))))
cm = lambda a: a.permute(*reversed(range(a.ndim))).contiguous().permute(*reversed(range(a.ndim)))

# ---- wsinterpd, kept dimensions only (ChannelData.sample / rectifyt0 at C1 size: T x N x M = 2048 x 64 x 32, one delay per (sample, transmit))
T, N, M = 2048, 64, 32
x = rn(T, N, M)
t = (torch.arange(T, device=dev, dtype=torch.float32).reshape(T, 1, 1) + torch.rand((1, 1, M), generator=g, device=dev) * 3).expand(T, 1, M).contiguous()
for interp in ("linear", "cubic"):
    ms = timed(lambda: wsinterpd(x, t, 1, 1, None, interp, 0.0))
    line(f"wsinterpd rectifyt0-like C1 {T}x{N}x{M} {interp}", ms, 2 * x.numel() * 8 + t.numel() * 4)
# ---- wsinterpd with a summed dimension (sample + sum over M: the general single-delay reduction); records in column-major order (and, second line, in torch order)
xc_, tc_ = cm(x), cm(t)
for interp in ("linear", "cubic"):
    ms = timed(lambda: wsinterpd(xc_, tc_, 1, 1, [3], interp, 0.0))
    line(f"wsinterpd sum over M  C1 {T}x{N}x{M} {interp}", ms, x.numel() * 8 + T * N * 8 + t.numel() * 4, f"{T * N * M / ms / 1e6:.1f} Gsample/s")
ms = timed(lambda: wsinterpd(x, t, 1, 1, [3], "cubic", 0.0))
line(f"  ... torch-order record (lanes along the sum) cubic", ms, x.numel() * 8 + T * N * 8 + t.numel() * 4, f"{T * N * M / ms / 1e6:.1f} Gsample/s")
# ---- C2-sized: T x N x M = 2048 x 128 x 128
T, N, M = 2048, 128, 128
x = rn(T, N, M)
t = (torch.arange(T, device=dev, dtype=torch.float32).reshape(T, 1, 1) + torch.rand((1, 1, M), generator=g, device=dev) * 3).expand(T, 1, M).contiguous()
ms = timed(lambda: wsinterpd(x, t, 1, 1, None, "cubic", 0.0))
line(f"wsinterpd rectifyt0-like C2 {T}x{N}x{M} cubic", ms, 2 * x.numel() * 8 + t.numel() * 4)
xc_, tc_ = cm(x), cm(t)
ms = timed(lambda: wsinterpd(xc_, tc_, 1, 1, [3], "cubic", 0.0))
line(f"wsinterpd sum over M  C2 {T}x{N}x{M} cubic", ms, x.numel() * 8 + T * N * 8 + t.numel() * 4, f"{T * N * M / ms / 1e6:.1f} Gsample/s")
ms = timed(lambda: wsinterpd(x, t, 1, 1, [3], "cubic", 0.0))
line(f"  ... torch-order record (lanes along the sum) cubic", ms, x.numel() * 8 + T * N * 8 + t.numel() * 4, f"{T * N * M / ms / 1e6:.1f} Gsample/s")
del xc_, tc_

# ---- focusTx at C1 (64-element FSA record -> 32 focused transmits): one split-delay launch per synthesised transmit, keep_rx
from qups_amd import ChannelData, Sequence, Transducer, UltrasoundSystem, Scan
from qups_amd.configs import workload
w = workload("c1")
T, N = w["T"], w["N"]
xf = rn(T, N, N)
xdc = Transducer(np.asarray(w["Pr"]), np.tile(np.array([[0.0], [0.0], [1.0]]), (1, N)))
foc = np.asarray(w["Pv"])[:3]
us = UltrasoundSystem(xdc, Sequence("FC", focus=foc, c0=w["c0"]), Scan(np.asarray(w["Pi"]).reshape(3, w["I1"], w["I2"], 1)))
chd = ChannelData(xf, 0.0, w["fs"])
try:
    ms = timed(lambda: us.focusTx(chd, interp="cubic"), reps=3)
    Mf = foc.shape[1]
    line(f"focusTx C1: {T}x{N}x{N} FSA -> {Mf} focused transmits, cubic", ms, xf.numel() * 8 + T * N * Mf * 8,
         f"{T * N * N * Mf / ms / 1e6:.1f} Gsample/s (every output sums {N} transmit elements)")
except Exception as ex:
    print("focusTx:", repr(ex))

# ---- hilbert (real fp32 traces -> analytic complex64): the one-pass LDS kernel and the hipFFT passes it replaces (QDAS_PRE_HIPFFT=1)
from qups_amd.preproc import hilbert
for name, (T, K) in {"C1 2048 x 64x32": (2048, 64 * 32), "C2 2048 x 128x128": (2048, 128 * 128), "C3 2816 x 256x256": (2816, 256 * 256), "1000 x 4096 (5^3 2^3)": (1000, 4096)}.items():
    xr = torch.randn((K, T), generator=g, device=dev, dtype=torch.float32).t()       # T x K, time fastest (the MATLAB memory order): no layout pass
    for path, env in (("one pass", "0"), ("hipFFT passes", "1")):
        os.environ["QDAS_PRE_HIPFFT"] = env
        try:
            ms = timed(lambda: hilbert(xr))
            line(f"hilbert {name} [{path}]", ms, xr.numel() * 4 + xr.numel() * 8, "(plan creation inside the call)")
        except Exception as ex:
            print("hilbert:", name, path, repr(ex))
    del xr
os.environ.pop("QDAS_PRE_HIPFFT", None)

# ---- convd: matched filter / ChannelData.filter along time, complex64 traces x a real 129-tap kernel, 'same'
from qups_amd.convd import convd
for name, (T, K) in {"C1 2048 x 2048": (2048, 2048), "C2 2048 x 16384": (2048, 16384), "C3 2816 x 65536": (2816, 65536)}.items():
    xc = rn(T, K)
    h = torch.randn((129, 1), generator=g, device=dev, dtype=torch.float32)
    try:
        ms = timed(lambda: convd(xc, h, 1, "same"))
        line(f"convd 'same' {name} * 129 taps", ms, 2 * xc.numel() * 8, f"{xc.numel() * 129 * 4 / ms / 1e9:.1f} TFLOP/s fp32 (complex x real tap = 2 FMA)")
    except Exception as ex:
        print("convd:", name, repr(ex))
    del xc
xc = rn(16384, 2048)                                                    # the same traces in MATLAB memory order (time fastest): the LDS-staged kernel
h = torch.randn((1, 129), generator=g, device=dev, dtype=torch.float32)
ms = timed(lambda: convd(xc, h, 2, "same"))
line("convd 'same' C2 2048 x 16384 * 129 taps, time fastest", ms, 2 * xc.numel() * 8, f"{xc.numel() * 129 * 4 / ms / 1e9:.1f} TFLOP/s fp32")
del xc

# ---- das_lut (bfDASLUT's kernel: delay tables instead of geometry) at C1 / C2 sizes, cubic, summed over receivers and transmits
for cfg in ("c1", "c2"):
    w = workload(cfg)
    T, N, M, I = w["T"], w["N"], w["M"], w["I1"] * w["I2"]
    xl = rn(T, N, M)
    Pi = torch.from_numpy(np.asarray(w["Pi"], np.float32).reshape(3, -1)).to(dev)
    Prt = torch.from_numpy(np.asarray(w["Pr"], np.float32)).to(dev)
    Pvt, Nvt = torch.from_numpy(np.asarray(w["Pv"], np.float32)).to(dev), torch.from_numpy(np.asarray(w["Nv"], np.float32)).to(dev)
    trx = (Pi.t().reshape(I, 1, 3) - Prt.t().reshape(1, N, 3)).norm(dim=2) * (w["fs"] / w["c0"])                 # the tables bfDASLUT builds: receive ...
    if cfg == "c2":
        ttx = (Pi.t() @ Nvt[:, :M]) * (w["fs"] / w["c0"]) - w["t0"] * w["fs"]                                   # ... and plane-wave transmit delays
    else:
        dv = Pi.t().reshape(I, 1, 3) - Pvt[:3].t().reshape(1, M, 3)
        ttx = (torch.sign(dv[..., 2]) * dv.norm(dim=2) + Pvt[2].reshape(1, M)) * (w["fs"] / w["c0"]) - w["t0"] * w["fs"]   # virtual sources
    trx, ttx = trx.reshape(w["I1"], w["I2"], N).contiguous(), ttx.reshape(w["I1"], w["I2"], M).contiguous()      # I1 x I2 x N: the image shape lets the fused tiled kernel take the call
    # the record and the tables in the REFERENCE's memory order (MATLAB column-major: fast time / depth contiguous -- what bfDASLUT hands wsinterpd2, kern/wsinterpd2.m:236):
    # views of buffers laid out the other way round, which the wrapper uses in place; torch-order tensors cost three layout copies per call (printed beside it)
    cm = lambda a: a.permute(*reversed(range(a.ndim))).contiguous().permute(*reversed(range(a.ndim)))
    xl_c, trx_c, ttx_c = cm(xl), cm(trx), cm(ttx)
    try:
        ms = timed(lambda: das_lut(xl_c, trx_c, ttx_c, interp="cubic"), reps=3)
        kern = getattr(das_lut, "last_kernel", "?")
        ms_t = timed(lambda: das_lut(xl, trx, ttx, interp="cubic"), reps=3)
        line(f"das_lut {cfg.upper()} I={I} N={N} M={M} cubic (geometric tables)", ms, xl.numel() * 8 + (trx.numel() + ttx.numel()) * 4 + I * 8, f"{I * N * M / ms / 1e6:.1f} Gpair/s  [{kern}]  (torch-order record and tables: {ms_t:.3f} ms)")
    except Exception as ex:
        print("das_lut:", cfg, repr(ex))
    del xl_c, trx_c, ttx_c
    del xl

# ---- greens: C1's simulator call (64 x 64 FSA, 2048 samples) with 1000 point scatterers
from qups_amd.greens import greens
w = workload("c1")
Pr = np.asarray(w["Pr"])
rng = np.random.default_rng(0)
for S in (1000, 100000):
    scat = np.stack([rng.uniform(-10e-3, 10e-3, S), np.zeros(S), rng.uniform(10e-3, 40e-3, S)])
    wv = np.hanning(33) * np.sin(2 * np.pi * 5e6 * np.arange(33) / (4 * w["fs"]))
    try:
        ms = timed(lambda: greens(Pr, Pr, scat, np.ones(S), w["c0"], wv, -16 / (4 * w["fs"]), 4 * w["fs"], w["fs"]), reps=3)
        y, _ = greens(Pr, Pr, scat, np.ones(S), w["c0"], wv, -16 / (4 * w["fs"]), 4 * w["fs"], w["fs"])
        line(f"greens C1 64x64 FSA, {S} scatterers, {y.shape[0]} samples", ms, y.numel() * 8, f"{S * y.shape[1] * y.shape[2] / ms / 1e6:.2f} G scatterer-paths/s (host marshalling inside the call)")
    except Exception as ex:
        print("greens:", S, repr(ex))
