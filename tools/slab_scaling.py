#!/usr/bin/env python3
"""Kernel time of ONE rank's pixel slab of the C3 frame for world sizes 1/2/4/8, measured on a single GPU (what each rank of `bench.py --gpus N` computes before
the all_gather: mirror slabs -- columns [c0, c1) of the first half and their mirror images in one plan -- on the hiprtc build).  NOT a scaling curve: no gather,
no xGMI, no second device.  Two layouts:
  equal width, every rank folds   -- rounds 3-4: `mirror_slab_columns`, each rank runs its own reciprocity fold of the replicated frame (a fixed ~0.4 ms)
  equal cost, prefolded frames    -- `balanced_column_bounds` over `measure_column_cost` (16 column blocks timed once), QDAS_PLAN_PREFOLDED plans fed the frame folded once
                                     by the acquisition rank (`bench.py --gpus N --prefolded`; the fold + replication is reported beside the timed region there)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qups_amd import DasPlan, build_problem, parse_options, _lib
from qups_amd.configs import workload
from qups_amd.dist import mirror_slab_columns, balanced_column_bounds, expand_block_cost, measure_column_cost

w = workload(sys.argv[1] if len(sys.argv) > 1 else "c3")
dev = torch.device("cuda:0")
T, N, M = w["T"], w["N"], w["M"]
I1, I2 = w["I1"], w["I2"]
g = torch.Generator(device=dev).manual_seed(1234)
xc = torch.view_as_complex(torch.randn((M, N, T, 2), generator=g, device=dev, dtype=torch.float32))
opts = parse_options(xc, list(w["opt"]) + ["interp", w["interp"], "input-precision", "single"])
prob = build_problem("DAS", w["Pi"], w["Pr"], w["Pv"], w["Nv"], (T, N, M), w["t0"], w["fs"], w["c0"], opts)
# the folded frame (what a PREFOLDED plan is handed)
xs = torch.zeros((M, N, T), dtype=torch.complex64, device=dev)
d = _lib.FoldDesc(T, N, 0, 0, 1, -1, None)
_lib.check(_lib.lib().qdas_fold(C.byref(d), C.c_void_p(xc.data_ptr()), C.c_void_p(xs.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
torch.cuda.synchronize()
cost, raw = measure_column_cost(prob, device=dev, nblocks=16, jit=True)
print("column-block kernel times [ms] of 16 equal blocks of the first half (outermost first), fold included:", " ".join(f"{v:.3f}" for v in raw))
print("cost profile (fixed part removed):", " ".join(f"{v:.3f}" for v in cost))


def time_slab(c0, c1, prefolded):
    kw = dict(prefolded=True) if prefolded else {}
    with DasPlan(prob, device=dev, i_begin=c0 * I1, i_count=(c1 - c0) * I1, jit=True, mirror_slab=True, **kw) as plan:
        plan.set_timing(True)
        k = []
        for _ in range(5):
            plan.execute_colmajor(xs if prefolded else xc, 1)
            k.append(plan.last_kernel_ms())
        return float(np.median(k[1:])), plan.aperture_split(), plan.tile_shape()


# (label, granule of the balanced boundaries -- 0: equal width --, prefolded frames)
for label, balanced, prefolded in (("equal width, every rank folds  ", 0, False), ("equal width, prefolded frames  ", 0, True), ("equal cost (32-column tiles), prefolded", 32, True),
                                   ("equal cost (any column), prefolded    ", 1, True)):
    base = None
    for world in (1, 2, 4, 8):
        bounds = balanced_column_bounds(expand_block_cost(cost, I2 // 2), world, balanced) if balanced else [mirror_slab_columns(I2, r, world)[0] for r in range(world)] + [I2 // 2]
        ts = []
        for rank in range(world):
            c0, c1 = bounds[rank], bounds[rank + 1]
            if world == 1:
                with DasPlan(prob, device=dev, jit=True, **(dict(prefolded=True) if prefolded else {})) as plan:
                    plan.set_timing(True)
                    k = []
                    for _ in range(5):
                        plan.execute_colmajor(xs if prefolded else xc, 1)
                        k.append(plan.last_kernel_ms())
                    ts.append((0, round(float(np.median(k[1:])), 3), plan.aperture_split(), c1 - c0))
            else:
                t, ks, _ = time_slab(c0, c1, prefolded)
                ts.append((rank, round(t, 3), ks, c1 - c0))
        worst = max(t[1] for t in ts)
        base = base or worst
        print(f"{label} world {world}: slowest rank {worst:.3f} ms (ideal {base / world:.3f}), efficiency {base / world / worst:.3f}   (rank, ms, workgroups per tile, columns): {ts}")
