#!/usr/bin/env python3
"""Plan-creation latency (upload, reciprocity test, probes) and host marshalling time of the BASELINE workloads."""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
from qups_amd import DasPlan, build_problem, parse_options
from qups_amd.configs import workload
for name in (sys.argv[1:] or ("c3", "c2", "c5", "c1", "c1f", "pw9")):
    w = workload(name)
    T, N, M = w["T"], w["N"], w["M"]
    xt = torch.zeros((2, 2, 2), dtype=torch.complex64)
    extra = ["interp", w["interp"], "input-precision", w["prec"]] + (["apod", w["apod"]] if w["apod"] is not None else [])
    t0 = time.perf_counter()
    prob = build_problem("DAS", w["Pi"], w["Pr"], w["Pv"], w["Nv"], (T, N, M), w["t0"], w["fs"], w["c0"], parse_options(xt, list(w["opt"]) + extra))
    t1 = time.perf_counter()
    for _ in range(3):
        torch.cuda.synchronize(); t2 = time.perf_counter()
        plan = DasPlan(prob)
        torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"{name}: host marshalling {1e3*(t1-t0):.1f} ms, plan creation (upload + reciprocity test + probes; prebuilt kernels) {1e3*(t3-t2):.1f} ms, kernel {plan.kernel}")
