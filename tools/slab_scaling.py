#!/usr/bin/env python3
"""Kernel time of ONE rank's pixel slab of the C3 frame for world sizes 1/2/4/8, measured on a single GPU
(what each rank of `bench.py --gpus N` computes before the all_gather)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qups_amd import DasPlan, build_problem, parse_options
from qups_amd.configs import workload

w = workload(sys.argv[1] if len(sys.argv) > 1 else "c3")
dev = torch.device("cuda:0")
T, N, M = w["T"], w["N"], w["M"]
I = w["I1"] * w["I2"]
g = torch.Generator(device=dev).manual_seed(1234)
xc = torch.view_as_complex(torch.randn((M, N, T, 2), generator=g, device=dev, dtype=torch.float32))
opts = parse_options(xc, list(w["opt"]) + ["interp", w["interp"], "input-precision", "single"])
prob = build_problem("DAS", w["Pi"], w["Pr"], w["Pv"], w["Nv"], (T, N, M), w["t0"], w["fs"], w["c0"], opts)
base = None
for world in (1, 2, 4, 8):
    ts = []
    for rank in sorted({0, world // 2, world - 1}):
        b, e = I * rank // world, I * (rank + 1) // world
        plan = DasPlan(prob, device=dev, i_begin=b, i_count=e - b, jit=bool(os.environ.get("QDAS_JIT")))
        plan.set_timing(True)
        k = []
        for _ in range(4):
            plan.execute_colmajor(xc, 1)
            k.append(plan.last_kernel_ms())
        ts.append((rank, float(np.mean(k[1:])), plan.aperture_split(), plan.tile_shape()))
    worst = max(t[1] for t in ts)
    base = base or worst
    print(f"world {world}: slowest rank {worst:.3f} ms (ideal {base / world:.3f}), efficiency {base / world / worst:.3f}", ts)
