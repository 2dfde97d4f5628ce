#!/usr/bin/env python3
"""bench.py -- headline benchmark of the DAS hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload c3|c2|small]

One "step" = one full-frame delay-and-sum of the workload (default: BASELINE config C3 -- 256-element
array, 256-transmit full-synthetic-aperture, 1024 x 1024 Cartesian scan, T = 2816, complex64 channel
data, lanczos3, no apodization, scalar sound speed) with the channel data
already resident in HBM.  With N > 1 (launched by torch.distributed.run, one rank per GPU) the SAME
image is split into N contiguous slabs of the linear pixel index, every rank beamforms its slab from
its own replica of the data, and the slabs are gathered with one RCCL all_gather -- inside the timed
region (strong scaling of one frame).

Rank 0 prints ONE JSON line: metric/value/unit per BASELINE.json plus
  "roofline":     algorithmic HBM bytes per launch / measured kernel time (hipEvents on the launch stream)
  "cpu_baseline": the C restatement of the reference's CPU branch (oracle/, kind "port") timed on this
                  host's cores on a pixel-subsampled image (N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3       # fp32 vector peak (same guide)
FLOP_PER_PAIR = {"nearest": 17, "linear": 24, "cubic": 42, "lanczos3": 54}   # SURVEY.md section 8d flop model


from qups_amd.configs import workload  # noqa: E402  (geometry of the BASELINE configs)


def cpu_baseline(w, x_host, budget_s=30.0):
    """Time the C oracle (port of the reference CPU branch) on a pixel-subsampled image."""
    from oracle import das_ref
    nthreads = das_ref.lib().das_ref_max_threads()

    def run(step):
        Pi = w["Pi"][:, ::step, ::step, :]
        t = time.perf_counter()
        das_ref.das_spec("DAS", Pi, w["Pr"], w["Pv"], w["Nv"], x_host, w["t0"], w["fs"], w["c0"],
                         VS="plane-waves" not in w["opt"], DV="diverging-waves" in w["opt"], interp=w["interp"],
                         prec="single", timing=True)
        return das_ref.LAST_SECONDS, Pi.shape[1] * Pi.shape[2]

    t, npx = run(32)                                   # calibration
    rate = npx / max(t, 1e-6)
    want = rate * budget_s
    step = int(np.clip(np.ceil(np.sqrt(w["I1"] * w["I2"] / max(want, 1.0))), 1, 32))
    t, npx = run(step)
    return {"value": round(npx / t / 1e6, 6), "unit": "Mpixel/s", "cores": int(nthreads), "kind": "port",
            "seconds": round(t, 3),
            "sample": f"every {step}th pixel per axis of the same image ({npx} px), full {w['N']}x{w['M']} aperture, "
                      f"float32, OpenMP x{nthreads}; full-frame time extrapolated: {w['I1'] * w['I2'] / (npx / t):.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 generic, 2 tiled")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--prec", default=None, help="override the workload's data precision (single | halfT); not the headline")
    ap.add_argument("--gen-apod", action="store_true", help="generate the workload's receive apodization inside the kernel "
                    "(qdas_desc.rx_apod_*) instead of streaming the materialised I x N array")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from qups_amd import DasPlan, build_problem, parse_options, _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, f"launch with torch.distributed.run --nproc-per-node {args.gpus}"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # QDAS_BENCH_SHARE_GPU=1 (plumbing check of the N > 1 path on a one-GPU box): every rank on cuda:0, gloo instead of RCCL
        share = os.environ.get("QDAS_BENCH_SHARE_GPU", "0") == "1"
        if share:
            local = 0
        torch.cuda.set_device(local)
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    else:
        torch.cuda.set_device(0)
    dev = torch.device(f"cuda:{local}")

    w = workload(args.workload)
    if args.prec:
        w["prec"] = args.prec
        w["label"] += f" [data precision {args.prec}]"
    T, N, M = w["T"], w["N"], w["M"]
    I = w["I1"] * w["I2"]
    g = torch.Generator(device=dev).manual_seed(1234)     # same data on every rank (replicated input)
    xc = torch.view_as_complex(torch.randn((M, N, T, 2), generator=g, device=dev, dtype=torch.float32))
    extra = ["interp", w["interp"], "input-precision", w["prec"]]
    if args.gen_apod and w["rx_apod"] is not None:
        from qups_amd.apodization import rx_apod_spec
        extra += ["rx-apod", rx_apod_spec(w["rx_apod"][0], normals=w["nrm"], **w["rx_apod"][1])]
        w["apod"] = None
    elif w["apod"] is not None:
        extra += ["apod", w["apod"]]
    opts = parse_options(xc, list(w["opt"]) + extra)
    if w["prec"] == "halfT":
        from qups_amd.das_spec import _cast_data
        xc = _cast_data(xc, "halfT", dev).contiguous()
    prob = build_problem("DAS", w["Pi"], w["Pr"], w["Pv"], w["Nv"], (T, N, M), w["t0"], w["fs"], w["c0"], opts)
    b, e = I * rank // world, I * (rank + 1) // world       # contiguous slab of the linear pixel index
    plan = DasPlan(prob, device=dev, kernel=args.kernel, i_begin=b, i_count=e - b)
    from qups_amd.dist import gather_pixels

    def step():
        y = plan.execute_colmajor(xc, 1)                       # (1, 1, 1, slab)
        if world > 1:
            y = gather_pixels(y, I, world)                     # one RCCL all_gather of the slabs -> (1, 1, 1, I) on every rank
        return y

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())

    # kernel-only time (hipEvents on the launch stream), outside the timed region
    plan.set_timing(True)
    kms = []
    for _ in range(max(2, min(args.steps, 5))):
        plan.execute_colmajor(xc, 1)
        kms.append(plan.last_kernel_ms())
    plan.set_timing(False)
    kernel_ms = float(np.mean(kms))
    fallback = plan.fallback_tiles()

    if rank == 0:
        ms = el / args.steps * 1e3
        pairs = I * N * M
        sb = 4 if w["prec"] == "halfT" else 8
        apb = 0 if w["apod"] is None else w["apod"].size * (2 if w["prec"] == "halfT" else 4)
        alg_bytes = (T * N * M * sb + 12 * I + sb * I + apb) / world      # per launch (per rank): x + Pi + y (+ apod)  (SURVEY 8d "B")
        traffic = None
        tfile = os.path.join(ROOT, "profiles", f"traffic_{w['name']}.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        info = _lib.device_info(local)
        rec = {
            "metric": "beamformed Mpixels/sec (1024^2 px, 256x256 Tx/Rx)" if w["name"] == "c3" else "beamformed Mpixels/sec",
            "value": round(I / (el / args.steps) / 1e6, 4), "unit": "Mpixel/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16" if w["prec"] == "halfT" else "f32", "data": "synthetic",
            "config": {"workload": w["label"], "pixels": I, "pairs_per_frame": pairs, "kernel": plan.kernel,
                       "fallback_tiles": fallback, "tile": list(plan.tile_shape()), "wave": list(plan.wave_shape()), "aperture_split": plan.aperture_split(), "parallelism": f"pixel-slab x{world} + RCCL all_gather" if world > 1 else "1 GPU",
                       "device": info["name"], "cu": info["cu_count"]},
            "roofline": {"bound": "hbm", "achieved": round(alg_bytes / (kernel_ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(alg_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                         "traffic": traffic, "kernel_ms": round(kernel_ms, 3), "algorithmic_bytes": int(alg_bytes),
                         "note": "compulsory-traffic accounting: this path is FP32-VALU / LDS-gather bound "
                                 "(~2.5e3 flop/byte), see also valu_frac",
                         "gpairs_per_s": round(pairs / world / (kernel_ms * 1e-3) / 1e9, 3),
                         "valu_tflops_model": round(pairs / world * FLOP_PER_PAIR.get(w["interp"], 42) / (kernel_ms * 1e-3) / 1e12, 3),
                         "valu_frac": round(pairs / world * FLOP_PER_PAIR.get(w["interp"], 42) / (kernel_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)},
        }
        if world == 1 and not args.no_cpu:
            try:
                xh = torch.view_as_real(xc).cpu().numpy().view(np.complex64).reshape(M, N, T).transpose(2, 1, 0)
                rec["cpu_baseline"] = cpu_baseline(w, xh)
            except Exception as ex:  # report, never hide
                rec["cpu_baseline"] = {"value": None, "unit": "Mpixel/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"FAILED: {ex!r}"}
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
