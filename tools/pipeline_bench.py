#!/usr/bin/env python3
"""PCIe-inclusive rate of the whole acquisition -> image chain on one MI355X for a stream of frames:

    pinned host int16 RF (T x N x M real) --H2D--> hilbert (pre.hip, one pass) -> [FIR band-pass (conv.hip)] -> DAS (tiled kernel)

Stage by stage (synchronised) and pipelined (uploads on a copy stream, double-buffered, overlapping the kernels of the previous
frame).  Usage: python tools/pipeline_bench.py [workload=c3] [frames=6] [taps=0]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qups_amd import DasPlan, _lib, build_problem, convd, parse_options
from qups_amd.configs import workload

w = workload(sys.argv[1] if len(sys.argv) > 1 else "c3")
F = int(sys.argv[2]) if len(sys.argv) > 2 else 6
taps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
T, N, M = w["T"], w["N"], w["M"]
K = N * M
dev = torch.device("cuda:0")
xt = torch.zeros((1, 1, 1), dtype=torch.complex64)
prob = build_problem("DAS", w["Pi"], w["Pr"], w["Pv"], w["Nv"], (T, N, M), w["t0"], w["fs"], w["c0"],
                     parse_options(xt, list(w["opt"]) + ["interp", w["interp"]]))
plan = DasPlan(prob, device=dev, jit=True)             # the kernel bench.py times (hiprtc build of the plan)
L = _lib.lib()
pd = _lib.PreDesc(T, K, T, _lib.QDAS_PRE_I16, 0, float(w["fs"]), 0.0, 0.0)
hp = C.c_void_p()
_lib.check(L.qdas_pre_plan_create(C.byref(hp), C.byref(pd)))
host = [torch.randint(-2000, 2000, (K, T), dtype=torch.int16).pin_memory() for _ in range(2)]
rf = [torch.empty((K, T), dtype=torch.int16, device=dev) for _ in range(2)]
xc = [torch.empty((M, N, T), dtype=torch.complex64, device=dev) for _ in range(2)]
h = None
if taps:
    k = np.arange(taps) - (taps - 1) / 2
    h = torch.from_numpy((np.sinc(0.75 * k) * 0.75 - np.sinc(0.25 * k) * 0.25) * np.hamming(taps)).to(dev, torch.complex64).reshape(1, 1, taps)
comp, copy = torch.cuda.Stream(), torch.cuda.Stream()


yout = torch.empty((1, 1, 1, prob.I), dtype=torch.complex64, device=dev)      # the image buffer of the stream (execute_into: no allocation per frame)


def kernels(b):
    _lib.check(L.qdas_pre_execute(hp, C.c_void_p(rf[b].data_ptr()), C.c_void_p(xc[b].data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    x = convd(xc[b], h, 3, "same") if taps else xc[b]
    return plan.execute_into(x, yout, 1)


def timed(fn, reps=3):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / reps


with torch.cuda.stream(comp):
    kernels(0); torch.cuda.synchronize()
    t_up = timed(lambda: rf[0].copy_(host[0], non_blocking=True))
    t_pre = timed(lambda: _lib.check(L.qdas_pre_execute(hp, C.c_void_p(rf[0].data_ptr()), C.c_void_p(xc[0].data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))))
    t_fir = timed(lambda: convd(xc[0], h, 3, "same")) if taps else 0.0
    t_das = timed(lambda: plan.execute_colmajor(xc[0], 1))
# pipelined stream of F frames
done = [torch.cuda.Event(), torch.cuda.Event()]
up = [torch.cuda.Event(), torch.cuda.Event()]
first_done, last_done = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for f in range(F):
    b = f & 1
    with torch.cuda.stream(copy):
        if f >= 2:
            copy.wait_event(done[b])
        rf[b].copy_(host[b], non_blocking=True)
        up[b].record(copy)
    with torch.cuda.stream(comp):
        comp.wait_event(up[b])
        y = kernels(b)
        done[b].record(comp)
        if f == 0:
            first_done.record(comp)
        if f == F - 1:
            last_done.record(comp)
torch.cuda.synchronize()
t_pipe = 1e3 * (time.perf_counter() - t0) / F
t_steady = first_done.elapsed_time(last_done) / (F - 1) if F > 1 else t_pipe      # frames 2..F: the first frame's upload has nothing to hide behind
I = prob.I
print(f"{w['name']}: {K * T * 2 / 1e9:.2f} GB int16 RF per frame (complex64 channel data would be {K * T * 8 / 1e9:.2f} GB)")
print(f"  stage by stage: upload {t_up:.2f} ms ({K * T * 2 / t_up * 1e-6:.1f} GB/s), hilbert {t_pre:.2f} ms, FIR({taps}) {t_fir:.2f} ms, DAS {t_das:.2f} ms, sum {t_up + t_pre + t_fir + t_das:.2f} ms")
print(f"  pipelined stream of {F} frames: {t_pipe:.2f} ms/frame = {I / t_pipe * 1e-3:.2f} Mpixel/s PCIe-inclusive (DAS alone: {I / t_das * 1e-3:.2f} Mpixel/s); "
      f"steady state (frames 2..{F}, device clock) {t_steady:.2f} ms/frame against max(upload, kernels) = {max(t_up, t_pre + t_fir + t_das):.2f}")
