"""convd on the C3 record (T = 2816 samples x 65536 traces, complex64, time contiguous, real taps): the direct kernel (csrc/conv.hip) against the FFT
convolution (csrc/pre.hip fftconv_launch) per filter length.  Usage: python tools/convd_fft_time.py [C2]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qups_amd import convd

T, K = (2048, 128 * 128) if sys.argv[1:] == ["C2"] else (2816, 256 * 256)
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.view_as_complex(torch.randn((K, T, 2), generator=g, device="cuda", dtype=torch.float32))


def timed(fn, R=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / R


print(f"# {K} traces x {T} samples complex64, real taps, 'same'; bytes = read + write once = {2 * T * K * 8 / 1e9:.2f} GB")
for taps in (16, 32, 48, 64, 96, 129, 192, 256, 512, 1024):
    h = torch.randn((1, taps), generator=g, device="cuda", dtype=torch.float32)
    os.environ["QDAS_CONV_FFT_MIN_TAPS"] = "1000000"
    md = timed(lambda: convd(x, h, 2, "same"))
    os.environ["QDAS_CONV_FFT_MIN_TAPS"] = "2"
    mf = timed(lambda: convd(x, h, 2, "same"))
    byts = 2 * T * K * 8
    print(f"taps={taps:5d}  direct {md:7.3f} ms ({T * K * taps * 4 / md * 1e-9:6.1f} TFLOP/s)   fft {mf:7.3f} ms ({byts / mf * 1e-6:7.1f} GB/s = {byts / mf * 1e-6 / 8000:.2f} of the HBM roof)   direct / fft = {md / mf:.2f}")
