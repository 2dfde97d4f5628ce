#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-resident path (QDAS_MEM_HOST, what the MEX shim uses): C3 frames staged from pageable host
memory -- one call per frame vs one call for the sequence (upload of frame f+1 overlaps the kernel of frame f)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qups_amd import _lib, build_problem, parse_options
from qups_amd.configs import workload

w = workload(sys.argv[1] if len(sys.argv) > 1 else "c3")
F = int(sys.argv[2]) if len(sys.argv) > 2 else 3
T, N, M = w["T"], w["N"], w["M"]
xt = torch.zeros((T, N, M), dtype=torch.complex64)
prob = build_problem("DAS", w["Pi"], w["Pr"], w["Pv"], w["Nv"], (T, N, M), w["t0"], w["fs"], w["c0"],
                     parse_options(xt, list(w["opt"]) + ["interp", w["interp"]]))
L = _lib.lib()
acs = (C.c_uint64 * len(prob.acstride))(*[int(v) for v in prob.acstride])
d = _lib.Desc()
d.sz = _lib.Sizes(T, N, M, prob.Isz[0], prob.Isz[1], prob.Isz[2], 0, prob.flag, int(prob.VS), int(prob.DV), _lib.QDAS_F32)
d.fs = prob.fs
keep = [np.ascontiguousarray(a) for a in (prob.Pi, prob.Pr, prob.Pv, prob.Nv, prob.cinv)]
d.Pi, d.Pr, d.Pv, d.Nv, d.cinv = (C.c_void_p(a.ctypes.data) for a in keep)
d.acstride, d.mem, d.device = acs, _lib.MEM_HOST, 0
h = C.c_void_p()
_lib.check(L.qdas_plan_create(C.byref(h), C.byref(d)))
rng = np.random.default_rng(0)
xh = rng.standard_normal((F, M, N, T, 2), dtype=np.float32)
I = prob.I
yh = np.zeros((F, I), np.complex64)
xp, yp = xh.ctypes.data, yh.ctypes.data
fb = T * N * M
for _ in range(2):
    t0 = time.perf_counter()
    for f in range(F):
        _lib.check(L.qdas_plan_execute(h, C.c_void_p(xp + f * fb * 8), C.c_void_p(yp + f * I * 8), None))
    t1 = time.perf_counter()
    _lib.check(L.qdas_plan_execute_frames(h, C.c_void_p(xp), C.c_void_p(yp), F, fb, I, None))
    t2 = time.perf_counter()
print(f"{w['name']}: {fb * 8 / 1e9:.2f} GB per frame from pageable host memory; frame by frame {1e3 * (t1 - t0) / F:.1f} ms/frame, "
      f"sequence of {F} (double-buffered) {1e3 * (t2 - t1) / F:.1f} ms/frame")
L.qdas_plan_destroy(h)
