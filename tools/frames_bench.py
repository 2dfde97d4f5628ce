#!/usr/bin/env python3
"""Frames streamed through one plan (the shape of the reference's own benchmark, test/ParTest.m:244-271: F = 10 frames):
ms per frame with frame pairs sharing a launch vs one launch per frame (QDAS_NO_FB2=1).  tools/frames_bench.py [workload] [F]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qups_amd import DasPlan, build_problem, parse_options
from qups_amd.configs import workload

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w = workload(name)
dev = torch.device("cuda:0")
T, N, M = w["T"], w["N"], w["M"]
g = torch.Generator(device=dev).manual_seed(1)
xc = torch.view_as_complex(torch.randn((F, M, N, T, 2), generator=g, device=dev, dtype=torch.float32))
extra = ["interp", w["interp"], "input-precision", w["prec"]] + (["apod", w["apod"]] if w["apod"] is not None else [])
opts = parse_options(xc, list(w["opt"]) + extra)
if w["prec"] == "halfT":
    from qups_amd.das_spec import _cast_data
    xc = _cast_data(xc, "halfT", dev).contiguous()
prob = build_problem("DAS", w["Pi"], w["Pr"], w["Pv"], w["Nv"], (T, N, M, F), w["t0"], w["fs"], w["c0"], opts)
modes = ("pairs", "pairs, hiprtc plan", "single", "single, hiprtc plan")
if name == "c3":          # reciprocal: the folded frame (two frames per launch / one) and, for reference, the unfolded reciprocal + mirror kernel
    modes += ("single, hiprtc plan, no fold",)
for mode in modes:
    os.environ.pop("QDAS_NO_FB2", None)
    if mode.startswith("single"):
        os.environ["QDAS_NO_FB2"] = "1"
    plan = DasPlan(prob, device=dev, jit="hiprtc" in mode, fold="no fold" not in mode)
    plan.set_timing(True)
    ms = []
    for _ in range(4):
        plan.execute_colmajor(xc, F)
        ms.append(plan.last_kernel_ms())
    print(f"{name} F={F} {mode:28s}: {np.mean(ms[1:]) / F:8.3f} ms/frame  ({w['I1'] * w['I2'] * F / np.mean(ms[1:]) / 1e3:8.1f} Mpixel/s)  {plan.kernel_name()}")
    plan.close()
