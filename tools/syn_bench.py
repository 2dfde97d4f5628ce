#!/usr/bin/env python3
"""'SYN' / 'MUL' modes (keep the receive / transmit dimension) on a BASELINE workload: tiled vs generic kernel, ms per frame.
tools/syn_bench.py [workload] [SYN|MUL]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qups_amd import DasPlan, build_problem, parse_options
from qups_amd.configs import workload

w = workload(sys.argv[1] if len(sys.argv) > 1 else "c2")
dev = torch.device("cuda:0")
T, N, M = w["T"], w["N"], w["M"]
g = torch.Generator(device=dev).manual_seed(1)
xc = torch.view_as_complex(torch.randn((M, N, T, 2), generator=g, device=dev, dtype=torch.float32))
opts = parse_options(xc, list(w["opt"]) + ["interp", w["interp"], "input-precision", "single"])
fun = sys.argv[2] if len(sys.argv) > 2 else "SYN"
prob = build_problem(fun, w["Pi"], w["Pr"], w["Pv"], w["Nv"], (T, N, M), w["t0"], w["fs"], w["c0"], opts)
for kern in (2, 1):
    plan = DasPlan(prob, device=dev, kernel=kern)
    plan.set_timing(True)
    ms = []
    for _ in range(3):
        y = plan.execute_colmajor(xc, 1)
        ms.append(plan.last_kernel_ms())
    print(f"{w['name']} {fun} {plan.kernel:8s}: {np.mean(ms[1:]):8.3f} ms  (output {tuple(y.shape)}, {y.numel() * 8 / 1e6:.0f} MB)")
