#!/usr/bin/env python3
"""The general (non-fused) kernels that ChannelData.sample / rectifyt0 / focusTx / bfDAS fall onto, timed on shapes of the BASELINE
configurations with a bytes-based roofline each:  tools/general_time.py  ->  one line per case:
   ms, algorithmic GB moved (inputs read once + outputs written once), GB/s, fraction of the 8 TB/s HBM roof.
Times are torch.cuda.Event pairs on torch's current stream -- the stream every call here launches on."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qups_amd.interpd import wsinterpd, das_lut

dev = torch.device("cuda:0")
HBM = 8000.0


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(np.median(ms))


def line(name, ms, nbytes, note=""):
    gbs = nbytes / ms / 1e6
    print(f"{name:58s} {ms:9.3f} ms  {nbytes / 1e9:7.3f} GB  {gbs:8.1f} GB/s  {gbs / HBM:6.3f} of HBM roof  {note}")


g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.view_as_complex(torch.randn(tuple(s) + (2,), generator=g, device=dev, dtype=torch.float32))

# ---- wsinterpd, kept dimensions only (ChannelData.sample / rectifyt0 at C1 size: T x N x M = 2048 x 64 x 32, one delay per (sample, transmit))
T, N, M = 2048, 64, 32
x = rn(T, N, M)
t = (torch.arange(T, device=dev, dtype=torch.float32).reshape(T, 1, 1) + torch.rand((1, 1, M), generator=g, device=dev) * 3).expand(T, 1, M).contiguous()
for interp in ("linear", "cubic"):
    ms = timed(lambda: wsinterpd(x, t, 1, 1, None, interp, 0.0))
    line(f"wsinterpd rectifyt0-like C1 {T}x{N}x{M} {interp}", ms, 2 * x.numel() * 8 + t.numel() * 4)
# ---- wsinterpd with a summed dimension (sample + sum over M: the general single-delay reduction)
for interp in ("linear", "cubic"):
    ms = timed(lambda: wsinterpd(x, t, 1, 1, [3], interp, 0.0))
    line(f"wsinterpd sum over M  C1 {T}x{N}x{M} {interp}", ms, x.numel() * 8 + T * N * 8 + t.numel() * 4, f"{T * N * M / ms / 1e6:.1f} Gsample/s")
# ---- C2-sized: T x N x M = 2048 x 128 x 128
T, N, M = 2048, 128, 128
x = rn(T, N, M)
t = (torch.arange(T, device=dev, dtype=torch.float32).reshape(T, 1, 1) + torch.rand((1, 1, M), generator=g, device=dev) * 3).expand(T, 1, M).contiguous()
ms = timed(lambda: wsinterpd(x, t, 1, 1, None, "cubic", 0.0))
line(f"wsinterpd rectifyt0-like C2 {T}x{N}x{M} cubic", ms, 2 * x.numel() * 8 + t.numel() * 4)
ms = timed(lambda: wsinterpd(x, t, 1, 1, [3], "cubic", 0.0))
line(f"wsinterpd sum over M  C2 {T}x{N}x{M} cubic", ms, x.numel() * 8 + T * N * 8 + t.numel() * 4, f"{T * N * M / ms / 1e6:.1f} Gsample/s")

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
