#!/usr/bin/env python3
"""BASELINE configuration C1 exactly as BASELINE.json names it -- 64-element linear array, 32 focused transmits, 256 x 256 ScanCartesian,
point scatterers through greens() -> focusTx -> bfDAS / DAS with linear interpolation -- timed call by call through the host mirror of the
reference's API (wall clock around synchronised calls: Python, marshalling, plan cache and kernels together).
Usage: python tools/c1_chain.py [scatterers=1000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qups_amd import ChannelData, Scan, Sequence, Transducer, UltrasoundSystem
from qups_amd.configs import workload

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
w = workload("c1")
fc, c0, fs, N = 5.208e6, w["c0"], w["fs"], w["N"]
xdc = Transducer(w["Pr"], w["nrm"], fc)
focus = w["Pv"][:3]
seq = Sequence("FC", focus=focus, c0=c0, numPulse=focus.shape[1])
us = UltrasoundSystem(xdc, seq, Scan(w["Pi"]), fs=fs)
rng = np.random.default_rng(0)
scat = np.stack([rng.uniform(-9e-3, 9e-3, S), np.zeros(S), rng.uniform(5e-3, 20e-3, S)])
t = np.arange(-2.0 / fc, 2.0 / fc, 1 / (4 * fs))
wv = np.exp(-(t * fc * 1.2) ** 2) * np.exp(2j * np.pi * fc * t)


def timed(fn, reps=5):
    out = fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return out, 1e3 * float(np.median(ts))


fsa, t_g = timed(lambda: us.greens(scat, np.ones(S), wv, t[0], 4 * fs, R0=c0 / fc, interp="linear", focus=False))
chd, t_f = timed(lambda: us.focusTx(fsa, seq, interp="linear"))
b1, t_l = timed(lambda: us.bfDAS(chd, interp="linear"))
b2, t_d = timed(lambda: us.DAS(chd, interp="linear"))
print(f"C1 as named, {S} scatterers: greens {tuple(fsa.data.shape)} {t_g:.2f} ms, focusTx -> {tuple(chd.data.shape)} {t_f:.2f} ms, "
      f"bfDAS (delay tables) {t_l:.2f} ms, DAS (fused kernel) {t_d:.2f} ms per call; |bfDAS - DAS| / max = "
      f"{float((b1 - b2).abs().max() / b2.abs().max()):.1e}")
