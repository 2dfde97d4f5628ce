#!/usr/bin/env python3
"""Kernel time of ONE rank's pixel slab of the C3 frame for world sizes 1/2/4/8, measured on a single GPU
(what each rank of `bench.py --gpus N` computes before the all_gather: mirror slabs -- columns [c0, c1) of the first half and their
mirror images in one plan -- on the hiprtc build).  NOT a scaling curve: no gather, no xGMI, no second device."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qups_amd import DasPlan, build_problem, parse_options
from qups_amd.configs import workload
from qups_amd.dist import mirror_slab_columns

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
        if world == 1:
            plan = DasPlan(prob, device=dev, jit=True)
        else:
            c0, c1 = mirror_slab_columns(w["I2"], rank, world)
            plan = DasPlan(prob, device=dev, i_begin=c0 * w["I1"], i_count=(c1 - c0) * w["I1"], jit=True, mirror_slab=True)
        plan.set_timing(True)
        k = []
        for _ in range(4):
            plan.execute_colmajor(xc, 1)
            k.append(plan.last_kernel_ms())
        ts.append((rank, round(float(np.mean(k[1:])), 3), plan.aperture_split(), plan.tile_shape()))
        plan.close()
    worst = max(t[1] for t in ts)
    base = base or worst
    print(f"world {world}: slowest rank {worst:.3f} ms (ideal {base / world:.3f}), efficiency {base / world / worst:.3f}", ts)
