#!/usr/bin/env python3
"""Time the point-scatterer simulator for a C3-shaped acquisition (256 x 256 traces, 2816 samples); with --stages, the impulse-train kernel's
stages at 100 000 scatterers (QDAS_GREENS_DBG cuts the kernel short: 4 after the trains are zeroed, 8 after scan + work-off, 16 after the conversion,
2 = scan without work-off)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qups_amd import geometry as G
from qups_amd.greens import greens_kernel

fc, c0 = 5e6, 1540.0
fs = 4 * fc
Pr = G.linear_array(256, 0.2e-3)[0]
t = np.arange(-2.5 / fc, 2.5 / fc, 1 / (4 * fs))
wv = (np.exp(-(t * fc * 1.2) ** 2) * np.exp(2j * np.pi * fc * t)).astype(np.complex64)
for I in (9, 1000, 100000):
    r = np.random.default_rng(0)
    Ps = np.stack([r.uniform(-30e-3, 30e-3, I), np.zeros(I), r.uniform(5e-3, 75e-3, I)])
    a = np.ones(I, np.complex64)
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = greens_kernel(Ps, a, Pr, Pr, wv, 2816, 0.0, float(t[0]), fs, 4.0, 1 / c0, c0 / fc, "cubic")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"I = {I:6d} scatterers: {dt * 1e3:8.2f} ms for 2816 x 256 x 256 samples ({I * 256 * 256 / dt / 1e9:.2f} G scatterer-traces/s)")

if "--stages" in sys.argv:
    def timed(dbg):
        if dbg: os.environ["QDAS_GREENS_DBG"] = str(dbg)
        else: os.environ.pop("QDAS_GREENS_DBG", None)
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            greens_kernel(Ps, a, Pr, Pr, wv, 2816, 0.0, float(t[0]), fs, 4.0, 1 / c0, c0 / fc, "cubic")
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        return dt * 1e3
    z, sc, so, cv, full = timed(4), timed(8), timed(2), timed(16), timed(0)
    print(f"stages at I = {I}: launch + sort + tables + zeroed trains {z:.2f} ms | scan + work-off {sc - z:.2f} (scan alone {so - z - (full - cv) - (cv - sc):.2f}) | "
          f"fixed point -> de-interleaved float {cv - sc:.2f} | convolution + store {full - cv:.2f} | total {full:.2f} ms")
