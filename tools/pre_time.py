#!/usr/bin/env python3
"""Time of hilbert (+ downmix) for a C3-shaped acquisition: 65536 real traces of 2816 samples."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np, torch
from qups_amd import _lib
dev = torch.device("cuda:0")
T, K = 2816, 65536
for N, typ, name in ((2816, _lib.QDAS_PRE_F32, "fp32 -> N=2816"), (4096, _lib.QDAS_PRE_I16, "int16 -> N=4096 + downmix")):
    x = (torch.randn((K, T), device=dev) * 1000).to(torch.int16 if typ == _lib.QDAS_PRE_I16 else torch.float32)
    y = torch.empty((K, N), dtype=torch.complex64, device=dev)
    d = _lib.PreDesc(T, K, N, typ, 0, 20e6, 0.0, 5e6 if typ == _lib.QDAS_PRE_I16 else 0.0)
    L = _lib.lib(); h = C.c_void_p()
    _lib.check(L.qdas_pre_plan_create(C.byref(h), C.byref(d)))
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _lib.check(L.qdas_pre_execute(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), None))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    L.qdas_pre_plan_destroy(h)
    print(f"hilbert {name}: {dt * 1e3:.2f} ms for {K} traces ({x.numel() * x.element_size() / 1e9:.2f} GB in, {y.numel() * 8 / 1e9:.2f} GB out)")
