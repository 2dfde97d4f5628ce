#!/usr/bin/env python3
"""End-to-end demo on one MI355X: simulate point targets (greens) -> take the real RF -> FIR band-pass (convd) -> hilbert on the device -> delay-and-sum
with an acceptance-angle apodization generated inside the kernel -> report where the image peaks.

    python examples/psf_demo.py

Mirrors the reference's example_.m flow (greens -> hilbert -> DAS) with this repository's device path only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from qups_amd import geometry as G, das_spec
from qups_amd.apodization import rx_apod_spec
from qups_amd.greens import greens
from qups_amd.preproc import hilbert
from qups_amd import convd


def main():
    fc, c0 = 5e6, 1500.0
    fs = 4 * fc
    Pr, nrm = G.linear_array(64, 0.3e-3)
    scat = np.array([[2e-3, -3e-3], [0.0, 0.0], [15e-3, 22e-3]])            # two point targets
    t = np.arange(-2.5 / fc, 2.5 / fc, 1 / (4 * fs))
    pulse = (np.exp(-(t * fc * 1.2) ** 2) * np.exp(2j * np.pi * fc * t)).astype(np.complex64)
    y, t0 = greens(Pr, Pr, scat, [1.0, 0.8], c0, pulse, float(t[0]), 4 * fs, fs, R0=c0 / fc, interp="cubic")   # S x N x M on the device
    rf = y.real.contiguous()                                                  # what a scanner delivers: real traces
    k = np.arange(63) - 31                                                    # zero-phase band-pass 0.5 fc .. 1.5 fc (windowed sinc)
    h = (2 * 1.5 * fc / fs * np.sinc(2 * 1.5 * fc / fs * k) - 2 * 0.5 * fc / fs * np.sinc(2 * 0.5 * fc / fs * k)) * np.hamming(63)
    rf = convd(rf, torch.from_numpy(h.astype(np.float32)).cuda(), 1, "same")    # along time (dim 1), every trace, on the device
    x = hilbert(rf)                                                           # analytic channel data, on the device
    xs = np.linspace(-8e-3, 8e-3, 257)
    zs = np.linspace(8e-3, 28e-3, 321)
    Pi = G.scan_cartesian(xs, zs)
    Pv, Nv, opt = G.sequence_args("FSA", tx_pos=Pr, tx_normals=nrm)
    b, plan = das_spec("DAS", Pi, Pr, Pv, Nv, x, t0, fs, c0, *opt, "interp", "cubic",
                       "rx-apod", rx_apod_spec("acceptance", theta=30.0, normals=nrm), return_plan=True)
    torch.cuda.synchronize()
    img = b.abs().cpu().numpy()[:, :, 0, 0, 0]
    peaks = []
    work = img.copy()
    for _ in range(2):
        iz, ix = np.unravel_index(np.argmax(work), work.shape)
        peaks.append((xs[ix] * 1e3, zs[iz] * 1e3, work[iz, ix]))
        work[max(0, iz - 20): iz + 21, max(0, ix - 20): ix + 21] = 0
    print(f"kernel {plan.kernel}, tile {plan.tile_shape()}, fallback tiles {plan.fallback_tiles()}")
    for (px, pz, v) in sorted(peaks):
        print(f"peak at x = {px:6.2f} mm, z = {pz:6.2f} mm, |b| = {v:.3g}")
    return sorted(peaks)


if __name__ == "__main__":
    main()
