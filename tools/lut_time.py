#!/usr/bin/env python3
"""Time of the split-delay flavour (qdas_das_lut, the bfDASLUT path) against the fused path on the same problem.
Usage: python tools/lut_time.py [workload=c2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qups_amd import das_lut, das_spec
from qups_amd.configs import workload

w = workload(sys.argv[1] if len(sys.argv) > 1 else "c2")
T, N, M = w["T"], w["N"], w["M"]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
x = torch.view_as_complex(torch.randn((T, N, M, 2), generator=g, device=dev, dtype=torch.float32))
Pi = torch.from_numpy(w["Pi"].reshape(3, -1)).to(dev)                     # 3 x I
Pr, Pv, Nv = (torch.from_numpy(np.asarray(a, np.float32)).to(dev) for a in (w["Pr"], w["Pv"], w["Nv"]))
I = Pi.shape[1]
cinv, fs, t0 = 1.0 / w["c0"], w["fs"], w["t0"]
# delay tables in samples, as bfDASLUT hands them over (src/UltrasoundSystem.m:4476-4673)
trx = (torch.linalg.norm(Pi[:, :, None] - Pr[:, None, :], dim=0) * cinv * fs).contiguous()                  # I x N
if "plane-waves" in w["opt"]:
    ttx = ((Pi[:, :, None] * Nv[:, None, :]).sum(0) * cinv - t0) * fs
else:
    dv = torch.linalg.norm(Pi[:, :, None] - Pv[:, None, :], dim=0)
    if "diverging-waves" not in w["opt"]:                 # focused: the distance is signed by the side of the focus (src/bf.cu:106-108)
        dv = torch.copysign(dv, ((Pi[:, :, None] - Pv[:, None, :]) * Nv[:, None, :]).sum(0))
    ttx = (dv * cinv - t0) * fs
ttx = ttx.contiguous()                                                                                   # I x M
# the image shape rides on the tables' leading dimensions (I1 x I2 x N, MATLAB order): the fused kernel needs depth-compact tiles
trx, ttx = trx.reshape(w["I1"], w["I2"], N), ttx.reshape(w["I1"], w["I2"], M)
for _ in range(2):
    y = das_lut(x, trx, ttx, interp=w["interp"])
torch.cuda.synchronize(); t = time.perf_counter()
R = 3
for _ in range(R):
    y = das_lut(x, trx, ttx, interp=w["interp"])
torch.cuda.synchronize(); ms_lut = 1e3 * (time.perf_counter() - t) / R
yf, plan = das_spec("DAS", w["Pi"], w["Pr"], w["Pv"], w["Nv"], x, t0, fs, w["c0"], *w["opt"], "interp", w["interp"], return_plan=True)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(R):
    plan.feval(x)
torch.cuda.synchronize(); ms_f = 1e3 * (time.perf_counter() - t) / R
a, b = y.reshape(-1), yf.reshape(-1)
err = float((a - b).abs().max() / b.abs().max())
print(f"{w['name']}: I = {I}, N x M = {N} x {M}, {w['interp']}: das_lut {ms_lut:.2f} ms ({I * N * M / ms_lut * 1e-9:.3f} Tpairs/s; tables {(N + M) * I * 4 / 1e9:.2f} GB), "
      f"fused das_spec {ms_f:.2f} ms incl. host-side layout copy; max |difference| / max = {err:.1e} (fp32 tables vs in-kernel delays)")
