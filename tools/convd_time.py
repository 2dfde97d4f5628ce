"""Time qdas_convd on channel-data-sized inputs (C3: T = 2816 samples x 65536 traces, complex64).  Usage: python tools/convd_time.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qups_amd import convd

T, K = 2816, 256 * 256
g = torch.Generator(device="cuda").manual_seed(0)
for cplx in (True, False):
    for taps in (32, 64, 256, 1024):
        for layout in ("time-contiguous (K x T)", "time-strided (T x K)"):
            dt = torch.complex64 if cplx else torch.float32
            if layout.startswith("time-c"):
                x = torch.randn((K, T), generator=g, device="cuda", dtype=torch.float32).to(dt)
                h = torch.randn((1, taps), generator=g, device="cuda", dtype=torch.float32).to(dt)
                dim = 2
            else:
                x = torch.randn((T, K), generator=g, device="cuda", dtype=torch.float32).to(dt)
                h = torch.randn((taps, 1), generator=g, device="cuda", dtype=torch.float32).to(dt)
                dim = 1
            for _ in range(2):
                convd(x, h, dim, "same")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            R = 5
            for _ in range(R):
                z = convd(x, h, dim, "same")
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / R
            flop = T * K * taps * (8 if cplx else 2)
            byts = 2 * T * K * (8 if cplx else 4)
            print(f"{'complex64' if cplx else 'float32  '} taps={taps:5d} {layout:26s} {ms:8.3f} ms  {flop / ms * 1e-9:8.1f} TFLOP/s  {byts / ms * 1e-6:8.1f} GB/s (read + write once)")
