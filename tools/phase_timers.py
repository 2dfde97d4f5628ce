#!/usr/bin/env python3
"""In-kernel phase timers of the tiled kernel (needs a -DQDAS_PROF=1 build: tools/abl/libqdas_prof.so).
Prints, per wave class (wave 0 / wave 15 of every workgroup), the mean time of each phase in s_memtime ticks and as a share
of the workgroup's lifetime."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("QDAS_LIB", os.path.join(ROOT, "tools/abl/libqdas_prof.so"))
import numpy as np, torch
from qups_amd import DasPlan, build_problem, parse_options, _lib
from qups_amd.configs import workload

w = workload(sys.argv[1] if len(sys.argv) > 1 else "c3")
F = int(sys.argv[2]) if len(sys.argv) > 2 else 1            # frames per call (2 / 4: shared launches)
dev = torch.device("cuda:0")
T, N, M = w["T"], w["N"], w["M"]
g = torch.Generator(device=dev).manual_seed(1234)
xc = torch.view_as_complex(torch.randn((F, M, N, T, 2), generator=g, device=dev, dtype=torch.float32))
prec = os.environ.get("QDAS_PT_PREC", w["prec"])                # the workload's data precision and pixel x receiver mask, as bench.py
extra = ["apod", w["apod"]] if w["apod"] is not None and not os.environ.get("QDAS_PT_NO_APOD") else []
opts = parse_options(xc, list(w["opt"]) + ["interp", w["interp"], "input-precision", prec] + extra)
if prec != "single":
    from qups_amd.das_spec import _cast_data
    xc = _cast_data(xc, prec, dev).contiguous()
prob = build_problem("DAS", w["Pi"], w["Pr"], w["Pv"], w["Nv"], (T, N, M, F), w["t0"], w["fs"], w["c0"], opts)
plan = DasPlan(prob, device=dev, reciprocal=not os.environ.get("QDAS_PT_NO_RECIPROCAL"))
plan.set_timing(True)
for _ in range(2):
    plan.execute_colmajor(xc, F)
ms = plan.last_kernel_ms()
torch.cuda.synchronize()
L = _lib.lib()
nwg = min(8192, (w["I1"] * w["I2"] + 1023) // 1024 * plan.aperture_split())
buf = np.zeros(nwg * 2 * 8, dtype=np.uint64)
L.qdas_debug_read_prof.argtypes = [C.c_void_p, C.c_size_t]
rc = L.qdas_debug_read_prof(buf.ctypes.data, buf.size)
assert rc == 0, rc
b = buf.reshape(nwg, 2, 8).astype(np.float64)
names = ["prologue", "wload+dma issue", "ra refresh", "pair loop", "wait+barrier", "workgroup total", "stages", "-"]
print(f"kernel {ms:.3f} ms, {nwg} workgroups, tile {plan.tile_shape()} wave {plan.wave_shape()}")
for wv, lab in ((0, "wave 0"), (1, "wave 15")):
    tot = b[:, wv, 5].mean()
    print(f"-- {lab}: workgroup lifetime {tot:.0f} ticks ({ms * 1e6 / (tot * nwg / 256):.2f} ns/tick if 256 workgroups run at a time)")
    for k in (0, 1, 2, 3, 4):
        print(f"   {names[k]:18s} {b[:, wv, k].mean():12.0f} ticks  {100 * b[:, wv, k].mean() / tot:5.1f} %   per stage {b[:, wv, k].mean() / b[:, wv, 6].mean():8.2f}")
    rest = tot - b[:, wv, :5].sum(1).mean()
    print(f"   {'unaccounted':18s} {rest:12.0f} ticks  {100 * rest / tot:5.1f} %")
# distribution of the workgroups' lifetimes (wave 0): how uneven are the tiles?  (a kernel with W workgroups per CU-slot ends when the slowest chain does)
tot0 = b[:, 0, 5]
q = np.percentile(tot0, [0, 5, 25, 50, 75, 95, 100])
print("workgroup lifetime percentiles (0/5/25/50/75/95/100):", " ".join(f"{v:.0f}" for v in q), f" max/mean {tot0.max() / tot0.mean():.3f}")
nz = plan.tile_shape()
if os.environ.get("QDAS_PT_DUMP"):
    np.save(os.environ["QDAS_PT_DUMP"], b)

