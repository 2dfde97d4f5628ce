#!/usr/bin/env python3
"""bench.py -- headline benchmark of the DAS hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload c3|c2|c5|c1|small]

One "step" = one full-frame delay-and-sum of the workload (default: BASELINE config C3 -- 256-element
array, 256-transmit full-synthetic-aperture, 1024 x 1024 Cartesian scan, T = 2816, complex64 channel
data, lanczos3, no apodization, scalar sound speed) with the channel data
already resident in HBM.

N > 1: one rank per GPU over RCCL.  Either launched by `python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...` (RANK / LOCAL_RANK / WORLD_SIZE in the environment) or -- when those are absent --
bench.py re-executes ITSELF under torch.distributed.run, so a bare `python bench.py --gpus 8` works too.
The SAME image is split into N contiguous slabs of the linear pixel index, every rank beamforms its slab
from its own replica of the data, and the slabs are gathered with one RCCL all_gather -- inside the timed
region (strong scaling of one frame).  Outside the timed region rank 0 also reports the per-rank kernel
times, the gather alone, and the cost of replicating the channel data from rank 0 (RCCL broadcast).

Rank 0 prints ONE JSON line: metric/value/unit per BASELINE.json plus
  "roofline":     algorithmic HBM bytes per launch / measured kernel time (hipEvents on the launch stream),
                  HBM traffic from rocprofv3 counter passes of this very command (N = 1; --traffic file|none to skip),
                  the fp32-VALU fraction on the model AND on the executed flop count of the kernel
  "general_ms_per_step": the same frame with the reciprocal mode disabled by a plan flag (what an apodized FSA costs)
  "cpu_baseline": the C restatement of the reference's CPU branch (oracle/) timed on this host's cores on a
                  pixel-subsampled image (N = 1 only): as written ("port") and rebuilt for this host ("port-tuned").
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3       # fp32 vector peak (same guide)
FLOP_PER_PAIR = {"nearest": 17, "linear": 24, "cubic": 42, "lanczos3": 54}   # SURVEY.md section 8d flop model
# Flops the tiled kernel's pair loop EXECUTES per (pixel, rx, tx) pair, counted from its instruction mix (DESIGN.md section 4.1:
# v_pk_add/mul = 2 flop, v_pk_fma = 4 flop per lane; index math 4 packed adds per transmit pair, weights, 4-tap complex MACs);
# reciprocal mode shares index + weights between the two traces of an unordered pair.  Cross-checked once against
# SQ_INSTS_VALU_{FMA,ADD,MUL}_F32 (profiles/r02/).  Per-stage code (receive delay in fp64, DMA issue) is not counted.
# The index + weight work is shared by the traces that have the same delay: 2 in reciprocal mode OR lateral-mirror mode, 4 in both.
EXEC_FLOP_GENERAL = {"nearest": 6.0, "linear": 14.0, "cubic": 37.0, "lanczos3": 53.0}      # one trace per delay
EXEC_FLOP_MAC = {"nearest": 4.0, "linear": 8.0, "cubic": 16.0, "lanczos3": 16.0}           # of which the complex multiply-accumulates (never shared)


def exec_flop_per_pair(interp, share, folded=False):
    """per (pixel, rx, tx) pair of the sum; a reciprocity-folded plan (fold.hip) forms ONE product per unordered pair on the pre-added traces:
    half the multiply-accumulates too (plus one add per sample and frame in the fold pass, not counted here)"""
    if interp not in EXEC_FLOP_GENERAL:
        return None
    f = EXEC_FLOP_MAC[interp] + (EXEC_FLOP_GENERAL[interp] - EXEC_FLOP_MAC[interp]) / share
    return f / 2 if folded else f


from qups_amd.configs import workload  # noqa: E402  (geometry of the BASELINE configs)


def host_info():
    model, phys = "?", None
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {l.split(":", 1)[0].strip(): l.split(":", 1)[1].strip() for l in out.splitlines() if ":" in l}
        model = kv.get("Model name", "?")
        phys = int(kv.get("Core(s) per socket", "0")) * int(kv.get("Socket(s)", "1")) or None
    except Exception:
        pass
    return model, phys


def cpu_baseline(w, x_host, budget_s=12.0, apod=None):
    """Time the C oracle (port of the reference CPU branch) on a pixel-subsampled image: the straight port and the
    host-tuned rebuild (-march=native, float weight math; oracle/Makefile).  Returns (records, lattice step, lattice image of the
    straight port) -- the image is what `parity_check` compares the GPU frame with."""
    from oracle import das_ref
    model, phys = host_info()
    out = []
    lattice = (None, None)
    for kind in ("port", "port-tuned"):
        try:
            L = das_ref.lib(tuned=(kind == "port-tuned"))
        except Exception as ex:
            out.append({"value": None, "unit": "Mpixel/s", "kind": kind, "sample": f"unavailable: {ex!r}"})
            continue
        nthreads = L.das_ref_max_threads()

        def run(step):
            Pi = w["Pi"][:, ::step, ::step, :]
            ap = () if apod is None else (apod[::step, ::step],)       # (pixel x receiver weights of the workload: C5's acceptance mask)
            img = das_ref.das_spec("DAS", Pi, w["Pr"], w["Pv"], w["Nv"], x_host, w["t0"], w["fs"], w["c0"],
                                   VS="plane-waves" not in w["opt"], DV="diverging-waves" in w["opt"], interp=w["interp"],
                                   apod=ap, prec="single", timing=True, tuned=(kind == "port-tuned"))
            return das_ref.LAST_SECONDS, Pi.shape[1] * Pi.shape[2], img

        try:
            t, npx, _ = run(32)                                # calibration
            rate = npx / max(t, 1e-6)
            want = rate * budget_s
            step = int(np.clip(np.ceil(np.sqrt(w["I1"] * w["I2"] / max(want, 1.0))), 1, 32))
            t, npx, img = run(step)
        except ValueError as ex:                               # (the tuned rebuild covers the unweighted fp32 sum only)
            out.append({"value": None, "unit": "Mpixel/s", "kind": kind, "sample": f"not applicable: {ex}"})
            continue
        if kind == "port":
            lattice = (step, img[:, :, 0, 0, 0])
        out.append({"value": round(npx / t / 1e6, 6), "unit": "Mpixel/s", "cores": int(nthreads), "kind": kind,
                    "seconds": round(t, 3), "cpu_model": model, "physical_cores": phys,
                    "gpairs_per_s": round(npx * w["N"] * w["M"] / t / 1e9, 4),
                    "sample": f"every {step}th pixel per axis of the same image ({npx} px), full {w['N']}x{w['M']} aperture, "
                              f"float32, OpenMP x{nthreads}; full-frame time extrapolated: {w['I1'] * w['I2'] / (npx / t):.1f} s"})
    return out, lattice[0], lattice[1]


def executed_pair_fraction(mask, wave):
    """Fraction of the (pixel, receiver) stages the fused kernel EXECUTES under a pixel x receiver weight `mask` (I1 x I2 x N, zero =
    no weight): a wave (wave[0] x wave[1] pixels) skips a receiver's stage only when all its 64 weights are zero (das_tile_impl.h),
    so the census is per wave footprint, not per pixel.  What `valu_frac_executed` scales the pair count with."""
    m = np.asarray(mask) != 0
    I1, I2, N = m.shape
    wz, wc = wave if wave and wave[0] else (1, 1)
    p1, p2 = (-I1) % wz, (-I2) % wc
    if p1 or p2:
        m = np.pad(m, ((0, p1), (0, p2), (0, 0)))
    blk = m.reshape((I1 + p1) // wz, wz, (I2 + p2) // wc, wc, N).any(axis=(1, 3))
    return float(blk.sum()) * wz * wc / float(I1 * I2 * N)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


SQ_PASS = ("GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_SALU")


def measure_traffic(argv):
    """Counters of the dominant kernel from rocprofv3 passes of THIS command (one pass per TCC counter: FETCH_SIZE and WRITE_SIZE do not
    fit together; one more pass for the clock / issue / LDS counters -- never combined with any tracing domain but --kernel-trace).
    HBM bytes per launch are corrected as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE (KiB) x 2 for 16 B/lane streaming
    reads, WRITE_SIZE (KiB) as is.  Returns (bytes | None, source string, {counter: average per full-frame dispatch, 'pass_kernel_ms': ...})."""
    import csv
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found", {}
    vals = {}
    tmp = tempfile.mkdtemp(prefix="qdas_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", QDAS_BENCH_CHILD="1")
    err = None
    try:
        for ctrs in (("FETCH_SIZE",), ("WRITE_SIZE",), SQ_PASS):
            d = os.path.join(tmp, ctrs[0])
            cmd = [exe, "--kernel-trace", "--pmc", *ctrs, "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__)] + argv + ["--steps", "2", "--warmup", "1", "--no-cpu", "--traffic", "none",
                                                                       "--no-general"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=420)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                err = f"rocprofv3 --pmc {' '.join(ctrs)} failed (rc {r.returncode})"
                if ctrs[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                    return None, err, {}
                break                                         # (the issue / LDS pass is an extra: the traffic figure stands without it)
            allrows = list(csv.DictReader(open(files[0])))
            tot = {}
            for x in allrows:                                # the dominant kernel = largest total duration (the frame kernel: prebuilt
                if x["Counter_Name"] == ctrs[0] and "issue_probe_kernel" not in x["Kernel_Name"]:      # das_tile_kernel<...> or hiprtc-built qdas_jit_tile__...)
                    tot[x["Kernel_Name"]] = tot.get(x["Kernel_Name"], 0) + int(x["End_Timestamp"]) - int(x["Start_Timestamp"])
            dom = max(tot, key=tot.get)
            rows = [x for x in allrows if x["Kernel_Name"] == dom]
            dmax = max(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in rows)
            rows = [x for x in rows if int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) >= 0.5 * dmax]   # (plan-time probe launches are short)
            for ctr in ctrs:
                full = [float(x["Counter_Value"]) for x in rows if x["Counter_Name"] == ctr]
                if full:
                    vals[ctr] = sum(full) / len(full)
            if ctrs[0] == "GRBM_GUI_ACTIVE":
                dur = [int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in rows if x["Counter_Name"] == ctrs[0]]
                vals["pass_kernel_ms"] = sum(dur) / len(dur) / 1e6
    except Exception as ex:
        if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
            return None, f"rocprofv3 counter pass failed: {ex!r}", {}
        err = f"issue / LDS counter pass failed: {ex!r}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    src = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of this command (FETCH_SIZE KiB x2 gfx950 correction + WRITE_SIZE KiB)"
    if err:
        src += "; " + err
    return int(vals["FETCH_SIZE"] * 2048 + vals["WRITE_SIZE"] * 1024), src, vals


def binding_roofs(vals, cu, pairs, taps, sample_bytes):
    """The roofs that actually bind this path, from the SQ / GRBM counters of the same run:
    effective_ghz = GRBM_GUI_ACTIVE / 8 XCDs / kernel time of the counter pass;  valu_issue_frac = wave64 VALU instructions x 4 cycles
    / (4 SIMDs x CUs) / active cycles;  lds_busy_frac = LDS-array cycles / CUs / active cycles;  lds_gather_floor_ms = the pair loop's tap
    bytes (pairs x taps x bytes per sample) at 256 B/clk/CU (MI355X_MICROARCH.md, LDS table: ds_read_b64) and the measured clock."""
    if not vals or "GRBM_GUI_ACTIVE" not in vals or not vals.get("pass_kernel_ms"):
        return {}
    act = vals["GRBM_GUI_ACTIVE"] / 8.0                        # active cycles of one XCD's clock domain
    ghz = act / (vals["pass_kernel_ms"] * 1e-3) / 1e9
    out = {"effective_ghz": round(ghz, 3), "counter_pass_kernel_ms": round(vals["pass_kernel_ms"], 3)}
    if "SQ_INSTS_VALU" in vals:
        out["valu_insts"] = int(vals["SQ_INSTS_VALU"])
        out["valu_issue_frac"] = round(vals["SQ_INSTS_VALU"] * 4.0 / (4 * cu) / act, 4)
        out["valu_insts_per_pair"] = round(vals["SQ_INSTS_VALU"] * 64.0 / pairs, 3)
    if "SQ_INSTS_SALU" in vals:
        out["salu_insts"] = int(vals["SQ_INSTS_SALU"])
    if "SQ_LDS_IDX_ACTIVE" in vals:
        out["lds_busy_frac"] = round(vals["SQ_LDS_IDX_ACTIVE"] / cu / act, 4)
        out["lds_bank_conflict_frac"] = round(vals.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(vals["SQ_LDS_IDX_ACTIVE"], 1.0), 5)
    if "SQ_INSTS_LDS" in vals:
        out["lds_insts"] = int(vals["SQ_INSTS_LDS"])
    out["lds_gather_bytes"] = int(pairs * taps * sample_bytes)
    out["lds_gather_floor_ms"] = round(pairs * taps * sample_bytes / (256.0 * cu * ghz * 1e9) * 1e3, 3)
    out["binding_note"] = ("VALU issue and the LDS gathers co-limit this kernel; the HBM frac above is the BASELINE's nominal roof, "
                           "unreachable at ~2.5e3 flop/byte")
    return out


def saturated_issue_rate(lib, cu):
    """csrc/probe.hip: ns per wave64 VALU instruction and SIMD at saturation for {broadcast pk_fma, v_fma_f32, the pair loop's VALU mix}, measured now, on this device."""
    import ctypes as C
    try:
        f = lib.qdas_debug_issue_rate
        f.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        ns = {}
        for name, mix in (("pk_fma_bcast", 0), ("fma_f32", 1), ("pair_mix", 2)):
            v, ms = C.c_double(), C.c_double()
            if f(-1, mix, C.byref(v), C.byref(ms)) != 0:
                return None
            ns[name] = round(v.value, 4)
        return {"ns": ns, "pair_mix_ginst_s": 4.0 * cu / ns["pair_mix"],
                "note": f"qdas_debug_issue_rate (csrc/probe.hip) in this run: {ns['pair_mix']} ns per wave64 VALU instruction and SIMD for the pair loop's VALU mix at 4 waves/SIMD on "
                        f"{cu} CUs ({ns['pk_fma_bcast']} ns broadcast v_pk_fma_f32, {ns['fma_f32']} ns v_fma_f32): peak = 4 SIMDs x CUs / ns"}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 generic, 2 tiled")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-general", action="store_true", help="skip the reciprocal-mode-off measurement")
    ap.add_argument("--no-reciprocal", action="store_true", help="disable the reciprocal mode for the headline measurement itself (plan flag)")
    ap.add_argument("--frames", type=int, default=1, help="frames per step: a STREAM of F distinct frames through the plan (the shape of the reference's own benchmark, "
                    "test/ParTest.m:244-271: F = 10): launches shared by two (four) frames compute tap index and weights once per group; value = I * F / t, "
                    "ms_per_frame reported; parity_check on the LAST frame of the last timed step; not the headline (default 1)")
    ap.add_argument("--no-fold", action="store_true", help="reciprocal plans: do not fold the frame (plan flag QDAS_PLAN_NO_FOLD): both traces of every unordered "
                    "transmit / receive pair are gathered per pixel, as in rounds 1-3")
    ap.add_argument("--no-jit", action="store_true", help="do not set the plan flag QDAS_PLAN_JIT: run the prebuilt instantiation instead of "
                    "the kernel hiprtc compiles for this plan's sizes (the reference's benchmark runs const-compiled kernels too: "
                    "src/UltrasoundSystem.m:5626-5748, test/ParTest.m:322-327)")
    ap.add_argument("--jit", action="store_true", help="(default; kept for compatibility)")
    ap.add_argument("--no-traffic", action="store_true", help="same as --traffic none")
    ap.add_argument("--checksum", action="store_true", help="add image_checksum (xxh3 / blake2b of the final image bytes on rank 0)")
    ap.add_argument("--traffic", default="auto", choices=["auto", "live", "file", "none"],
                    help="roofline.traffic: rocprofv3 counter passes of this command (live; auto = live at N=1) or profiles/traffic_<w>.json")
    ap.add_argument("--fmod", type=float, default=0.0, help="remodulation frequency [Hz] ('modulation' option): baseband data; not the headline")
    ap.add_argument("--rx-apod", default=None, help="generated receive apodization for ANY workload, e.g. fnumber:1.5 | acceptance:30 | cosine:45 "
                    "(evaluated inside the kernel: what an apodized frame costs); not the headline")
    ap.add_argument("--window-apod", action="store_true", help="Hann receive window x Hann transmit window (pixel-independent apodization, folded "
                    "into an N x M table): an apodized full-synthetic-aperture frame stays in the reciprocal mode; not the headline")
    ap.add_argument("--tx-apod", default=None, choices=["multiline", "scanline"], help="transmit-side rule of a focused workload (c1, c1f) in the reference's "
                    "shape 1 x I2 x 1 x 1 x M (UltrasoundSystem.apMultiline / apScanline); not the headline")
    ap.add_argument("--rx-apod-array", action="store_true", help="with --rx-apod: pass the MATERIALISED I x N array instead of the in-kernel rule "
                    "(with --tx-apod: a transmit-side rule and a receive-side mask together -- per-pair pixel weights)")
    ap.add_argument("--prec", default=None, help="override the workload's data precision (single | halfT | double); not the headline")
    ap.add_argument("--balance", action="store_true", help="--gpus N > 1: mirror slabs of equal measured COST instead of equal width (rank 0 times the column blocks of the first "
                    "half once at plan creation, broadcasts the profile; boundaries in multiples of the 32-column tiles: qups_amd.dist balanced_column_bounds).  Off by default: "
                    "at C3 the ranks' slabs are one tile round each and whole tiles are too coarse to trade (profiles/r05/slab_kernel_times_c3.txt)")
    ap.add_argument("--prefolded", action="store_true", help="--gpus N > 1, reciprocal fp32 workloads: the timed region is handed FOLDED frames (folded once per acquisition by rank 0 "
                    "and replicated as the packed upper triangle, outside the timed region; QDAS_PLAN_PREFOLDED plans: no per-rank fold pass -- the fixed ~0.4 ms per rank that "
                    "does not shrink with the slab).  Reported as such (config.prefolded_timed_region), with the fold + replication time beside it; not the default")
    ap.add_argument("--gen-apod", action="store_true", help="generate the workload's receive apodization inside the kernel "
                    "(qdas_desc.rx_apod_*) instead of streaming the materialised I x N array")
    args = ap.parse_args()
    args.jit = not args.no_jit
    if args.no_traffic:
        args.traffic = "none"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist
    from qups_amd import DasPlan, build_problem, parse_options, _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = None
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # QDAS_BENCH_SHARE_GPU=1 (plumbing check of the N > 1 path on a one-GPU box): every rank on cuda:0, gloo instead of RCCL
        share = os.environ.get("QDAS_BENCH_SHARE_GPU", "0") == "1"
        if not share and torch.cuda.device_count() < world:
            raise SystemExit(f"bench.py --gpus {world}: only {torch.cuda.device_count()} HIP device(s) visible")
        if share:
            local = 0
        torch.cuda.set_device(local)
        backend = "gloo" if share else "nccl"
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
    F = max(1, args.frames)
    xc = torch.view_as_complex(torch.randn((F, M, N, T, 2), generator=g, device=dev, dtype=torch.float32))
    if F == 1:
        xc = xc[0]
    extra = ["interp", w["interp"], "input-precision", w["prec"]] + (["modulation", args.fmod] if args.fmod else [])
    if args.fmod:
        w["label"] += f" [fmod {args.fmod:g} Hz]"
    if args.window_apod:
        extra += ["apod", np.hanning(N + 2)[1:-1].astype(np.float32).reshape(1, 1, 1, N, 1), "apod", np.hanning(M + 2)[1:-1].astype(np.float32).reshape(1, 1, 1, 1, M)]
        w["label"] += " [Hann receive x transmit windows]"
    if args.tx_apod:
        from qups_amd import apodization as A
        Pi3 = np.asarray(w["Pi"]).reshape(3, w["I1"], w["I2"], -1)
        xi, xv = Pi3[0, 0, :, 0], np.asarray(w["Pv"])[0]
        a = A.ap_multiline(xi, xv) if args.tx_apod == "multiline" else A.ap_scanline(xi, xv)
        extra += ["apod", a.astype(np.float32)]
        w["label"] += f" [{args.tx_apod} transmit apodization]"
    if args.rx_apod:
        from qups_amd.apodization import rx_apod_spec
        kind, _, par = args.rx_apod.partition(":")
        kw = ({"f": float(par)} if kind == "fnumber" else {"theta": float(par)}) if par else {}
        if args.rx_apod_array:
            from qups_amd import apodization as A
            fn = {"acceptance": A.ap_acceptance_angle, "cosine": A.ap_cosine_angle, "fnumber": A.ap_aperture_growth}[kind]
            nrm = w.get("nrm")
            if nrm is None:
                nrm = np.tile(np.array([[0.0], [0.0], [1.0]]), (1, N))
            extra += ["apod", fn(w["Pi"], w["Pr"], nrm, *kw.values()).astype(np.float32)]
            w["label"] += f" [receive apodization array {args.rx_apod}]"
        else:
            extra += ["rx-apod", rx_apod_spec(kind, normals=w.get("nrm"), **kw)]
            w["label"] += f" [generated receive apodization {args.rx_apod}]"
        w["apod"] = None
    elif args.gen_apod and w["rx_apod"] is not None:
        from qups_amd.apodization import rx_apod_spec
        extra += ["rx-apod", rx_apod_spec(w["rx_apod"][0], normals=w["nrm"], **w["rx_apod"][1])]
        w["apod"] = None
    elif w["apod"] is not None:
        extra += ["apod", w["apod"]]
    opts = parse_options(xc, list(w["opt"]) + extra)
    if w["prec"] != "single":
        from qups_amd.das_spec import _cast_data
        xc = _cast_data(xc, w["prec"], dev).contiguous()
    prob = build_problem("DAS", w["Pi"], w["Pr"], w["Pv"], w["Nv"], (T, N, M), w["t0"], w["fs"], w["c0"], opts)
    # N > 1: qups_amd.dist -- pixel slabs (mirror slabs when every rank's plan takes the lateral-mirror mode: rank r beamforms columns of the
    # first half AND their mirror images) and ONE RCCL all_gather
    from qups_amd.dist import ShardedDasPlan
    balance = "measure" if (world > 1 and args.balance) else None
    prefolded_run = bool(args.prefolded and world > 1 and w["prec"] == "single" and not args.no_fold and not args.no_reciprocal and F == 1)
    splan = ShardedDasPlan(prob, rank, world, device=dev, kernel=args.kernel, reciprocal=not args.no_reciprocal, jit=args.jit, fold=not args.no_fold, balance=balance,
                           **({"prefolded": True} if prefolded_run else {}))
    plan = splan.plan
    prefold_ms = None
    if prefolded_run:                                   # the acquisition rank folds ONCE, the packed upper triangle travels, every rank unpacks: outside the timed region, timed alone
        from qups_amd.dist import FoldedReplicator
        rep0 = FoldedReplicator(N, T, dev, src=0)
        slot0, work0 = rep0.send(xc if rank == 0 else None, rank, async_op=True)
        xc_unfolded, xc = xc, rep0.receive(slot0, rank, work0)
        torch.cuda.synchronize(); dist.barrier()
        tpf = time.perf_counter()
        slot0, work0 = rep0.send(xc_unfolded if rank == 0 else None, rank, async_op=True)
        xc = rep0.receive(slot0, rank, work0)
        torch.cuda.synchronize(); dist.barrier()
        prefold_ms = (time.perf_counter() - tpf) * 1e3
    b, e = splan.i_begin, splan.i_begin + splan.i_count
    slab_kw = dict(i_begin=b, i_count=e - b, mirror_slab=splan.mirror_slabs)

    yslab = torch.empty((F, 1, 1, splan.out_count), dtype=xc.dtype, device=dev)      # the image buffer of the frame stream (reused: execute_into)

    def step():
        y = plan.execute_into(xc, yslab, F)                    # (F, 1, 1, slab)
        if world > 1:
            y = splan.gather(y)                                # one RCCL all_gather of the slabs -> (F, 1, 1, I) on every rank
        return y

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        yimg = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    yimg = yimg.clone()                                        # the frame of the last timed step (checksum, parity_check)
    if world > 1:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())

    # ---- outside the timed region
    def kernel_time(p, reps):
        p.set_timing(True)
        ks = []
        for _ in range(reps):
            p.execute_into(xc, yslab, F)
            ks.append(p.last_kernel_ms() / F)                  # (per frame)
        p.set_timing(False)
        return float(np.mean(ks))

    kernel_ms = kernel_time(plan, max(2, min(args.steps, 5)))     # hipEvents on the launch stream
    fallback = plan.fallback_tiles()
    multi = {}
    if world > 1:
        km = torch.tensor([kernel_ms], device=dev, dtype=torch.float64)
        allk = [torch.zeros_like(km) for _ in range(world)]
        dist.all_gather(allk, km)
        # what every rank's plan looks like (the first multi-GPU run should say where the time goes): slab, tile / wave footprint,
        # workgroups per tile, modes -- gathered as one small integer tensor per rank
        tz, tcx = plan.tile_shape()
        wz, wc = plan.wave_shape()
        info_t = torch.tensor([rank, b, e - b, tz, tcx, wz, wc, plan.aperture_split(), int(plan.mirror), int(plan.folded), int(plan.reciprocal), plan.fallback_tiles()],
                              device=dev, dtype=torch.int64)
        infos = [torch.zeros_like(info_t) for _ in range(world)]
        dist.all_gather(infos, info_t)
        y = plan.execute_colmajor(xc, F)
        torch.cuda.synchronize(); dist.barrier()
        tg = time.perf_counter()
        for _ in range(5):
            splan.gather(y)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) / 5 * 1e3
        # replicating the channel data from rank 0 (what a single acquisition host has to do once per frame): one RCCL broadcast
        xr = torch.view_as_real(xc) if xc.is_complex() else xc
        dist.broadcast(xr, 0)                                   # warm-up (connection setup)
        torch.cuda.synchronize(); dist.barrier()
        tb = time.perf_counter()
        for _ in range(3):
            dist.broadcast(xr, 0)
        torch.cuda.synchronize(); dist.barrier()
        bcast_ms = (time.perf_counter() - tb) / 3 * 1e3
        # a STREAM of frames from one acquisition rank: the replication of frame f+1 (asynchronous broadcast into a second buffer) overlaps the
        # beamforming + gather of frame f -- what the replication really costs a pipeline; 4 frames, first one not counted
        stream_ms = None
        try:
            xa, xb = xr, torch.empty_like(xr)
            bufs = [xa, xb]
            work = dist.broadcast(bufs[1], 0, async_op=True)
            torch.cuda.synchronize(); dist.barrier()
            ts = time.perf_counter()
            nfs = 4
            for f in range(nfs):
                cur, nxt = bufs[(f + 1) % 2], bufs[f % 2]
                work.wait()                                        # frame f has arrived in `cur`
                if f + 1 < nfs:
                    work = dist.broadcast(nxt, 0, async_op=True)   # frame f+1 travels while frame f is beamformed
                curc = torch.view_as_complex(cur) if xc.is_complex() and not cur.is_complex() else cur
                yb = plan.execute_into(curc.reshape(xc.shape), yslab, F)
                splan.gather(yb)
            torch.cuda.synchronize(); dist.barrier()
            stream_ms = (time.perf_counter() - ts) / nfs * 1e3
        except Exception as ex:                                    # (reported, never fatal: the headline above stands on its own)
            stream_ms = f"failed: {ex!r}"
        # the same stream with the frame travelling FOLDED (reciprocal acquisitions: qups_amd.dist.FoldedReplicator -- rank 0 folds once, ONE broadcast of the
        # packed upper triangle = half the bytes, every rank beamforms the folded frame on a QDAS_PLAN_PREFOLDED plan: no per-rank fold pass either)
        folded_stream = None
        if bool(plan.folded) and F == 1 and w["prec"] == "single" and not args.no_fold and not prefolded_run:      # (--prefolded: the headline itself ran that way)
            try:
                from qups_amd.dist import FoldedReplicator
                fsplan = ShardedDasPlan(prob, rank, world, device=dev, kernel=args.kernel, reciprocal=True, jit=args.jit, prefolded=True,
                                        mirror_slabs=splan.mirror_slabs)
                rep = FoldedReplicator(N, T, dev, src=0)
                yf = torch.empty((1, 1, 1, fsplan.out_count), dtype=xc.dtype, device=dev)
                slot, work = rep.send(xc if rank == 0 else None, rank, async_op=True)
                xs = rep.receive(slot, rank, work)
                ychk = fsplan.gather(fsplan.plan.execute_into(xs, yf, 1)).reshape(-1)       # (also: warm-up)
                fold_err = float((torch.view_as_real(ychk) - torch.view_as_real(yimg.reshape(-1))).abs().max() / torch.view_as_real(yimg).abs().max())
                slot, work = rep.send(xc if rank == 0 else None, rank, async_op=True)
                torch.cuda.synchronize(); dist.barrier()
                ts = time.perf_counter()
                nfs = 4
                for f in range(nfs):
                    xs = rep.receive(slot, rank, work)                 # frame f has arrived (unpacked on the receiving ranks)
                    if f + 1 < nfs:
                        slot, work = rep.send(xc if rank == 0 else None, rank, async_op=True)   # frame f + 1: fold + pack on rank 0, then the broadcast
                    fsplan.gather(fsplan.plan.execute_into(xs, yf, 1))
                torch.cuda.synchronize(); dist.barrier()
                folded_stream = {"ms_per_step": round((time.perf_counter() - ts) / nfs * 1e3, 3), "bytes_per_frame": rep.bytes_per_frame,
                                 "image_vs_headline": float(f"{fold_err:.2e}"),
                                 "note": "rank 0 folds each frame once (qdas_fold), ONE broadcast of the packed upper triangle (half the frame), PREFOLDED plans on every rank"}
                fsplan.close()
            except Exception as ex:
                folded_stream = {"ms_per_step": None, "note": f"failed: {ex!r}"}
        # the OTHER slab layouts beside the one the timed region ran (the first run on real hardware uses the driver's flag-less command and should
        # yield all of them): the same barrier-bracketed loop over `steps` steps, max over ranks, outside the headline's timed region.  Reciprocal fp32 frames only.
        layouts = {}
        this_layout = ("equal_cost_" if splan.col_bounds is not None else "equal_width_") + ("prefolded" if prefolded_run else "each_rank_folds" if bool(plan.folded) else "no_fold")
        layouts[this_layout] = {"ms_per_step": round(el / args.steps * 1e3, 3), "timed_region": True}
        if bool(plan.folded) and F == 1 and w["prec"] == "single" and not args.no_fold and not args.no_reciprocal:
            from qups_amd.dist import FoldedReplicator
            x_unf = xc_unfolded if prefolded_run else xc
            xfold = xc if prefolded_run else None
            for name, kw in (("equal_width_each_rank_folds", dict()), ("equal_width_prefolded", dict(prefolded=True)), ("equal_cost_prefolded", dict(prefolded=True, balance="measure"))):
                if name in layouts:
                    continue
                # (a plan that one rank alone fails to make -- a compiler cache race, memory -- must not leave the others inside a collective: the ranks agree first)
                sp2, why = None, ""
                try:
                    sp2 = ShardedDasPlan(prob, rank, world, device=dev, kernel=args.kernel, reciprocal=True, jit=args.jit, **kw)
                except Exception as ex:
                    why = repr(ex)
                okf = torch.tensor([1 if sp2 is not None else 0], device=dev, dtype=torch.int32)
                dist.all_reduce(okf, op=dist.ReduceOp.MIN)
                if int(okf.item()) == 0:
                    layouts[name] = {"ms_per_step": None, "note": "failed: plan creation on some rank" + (f" (this rank: {why})" if why else "")}
                    if sp2 is not None:
                        sp2.close()
                    continue
                try:
                    if kw.get("prefolded") and xfold is None:
                        rp = FoldedReplicator(N, T, dev, src=0)
                        sl, wk = rp.send(x_unf if rank == 0 else None, rank, async_op=True)
                        xfold = rp.receive(sl, rank, wk).clone()
                    xin = xfold if kw.get("prefolded") else x_unf
                    y2 = torch.empty((F, 1, 1, sp2.out_count), dtype=xc.dtype, device=dev)
                    for _ in range(max(1, args.warmup)):
                        sp2.gather(sp2.plan.execute_into(xin, y2, F))
                    torch.cuda.synchronize(); dist.barrier()
                    t2 = time.perf_counter()
                    for _ in range(args.steps):
                        yl = sp2.gather(sp2.plan.execute_into(xin, y2, F))
                    torch.cuda.synchronize(); dist.barrier()
                    tl = torch.tensor([time.perf_counter() - t2], device=dev, dtype=torch.float64)
                    dist.all_reduce(tl, op=dist.ReduceOp.MAX)
                    err2 = float((torch.view_as_real(yl.reshape(-1)) - torch.view_as_real(yimg.reshape(-1))).abs().max() / torch.view_as_real(yimg).abs().max())
                    layouts[name] = {"ms_per_step": round(float(tl.item()) / args.steps * 1e3, 3), "timed_region": False, "image_vs_headline": float(f"{err2:.2e}"),
                                     "slab_columns": None if sp2.col_bounds is None else [int(v) for v in sp2.col_bounds]}
                    sp2.close()
                except Exception as ex:
                    layouts[name] = {"ms_per_step": None, "note": f"failed: {ex!r}"}
        seen = [None] * world
        try:                                            # who is in the process group: rank, local device ordinal, device name, PCI bus id (one tiny object gather)
            props = torch.cuda.get_device_properties(dev)
            dist.all_gather_object(seen, {"rank": rank, "local_device": int(dev.index if dev.index is not None else local), "name": props.name,
                                          "pci": getattr(props, "pci_bus_id", None), "host": os.uname().nodename})
        except Exception as ex:
            seen = [f"all_gather_object failed: {ex!r}"]
        multi = {"backend": backend, "rccl_ranks": dist.get_world_size() if backend == "nccl" else 0, "ranks_seen": seen,
                 "slab_columns": None if splan.col_bounds is None else [int(v) for v in splan.col_bounds],
                 "slab_layout": ("mirror slabs of equal measured cost (rank 0's per-block kernel times, broadcast)" if splan.col_bounds is not None else
                                 "mirror slabs of equal width" if splan.mirror_slabs else "contiguous pixel slabs"),
                 "layout": this_layout, "layouts": layouts,
                 "layouts_note": "layout = what the timed region ran; layouts = every slab layout measured in this invocation with the same loop (prefolded: rank 0 folds the frame ONCE "
                                 "and replicates the packed upper triangle outside the loop -- fold_and_replicate_ms; equal_cost: column ranges of equal measured cost, in whole tiles); one "
                                 "GPU timing every rank's slab predicted 74 / 93 / 92 % at 8 ranks for the three (profiles/r05/slab_kernel_times_c3.txt): NOT a measured curve",
                 "prefolded_timed_region": prefolded_run, "fold_and_replicate_ms": None if prefold_ms is None else round(prefold_ms, 3),
                 "stream_folded_replication": folded_stream,
                 "stream_ms_per_step_incl_overlapped_replication": round(stream_ms, 3) if isinstance(stream_ms, float) else stream_ms,
                 "per_rank_kernel_ms": [round(float(k.item()), 3) for k in allk], "gather_ms": round(gather_ms, 3),
                 "slowest_rank": int(np.argmax([float(k.item()) for k in allk])),
                 "kernel_balance": round(float(np.mean([float(k.item()) for k in allk]) / max(float(k.item()) for k in allk)), 4),
                 "per_rank_plan": [dict(zip(("rank", "i_begin", "i_count", "tile_z", "tile_cols", "wave_z", "wave_cols", "aperture_split", "mirror", "folded", "reciprocal", "fallback_tiles"),
                                            [int(v) for v in it.tolist()])) for it in infos],
                 "note": "per_rank_kernel_ms includes each rank's own reciprocity fold of the replicated frame (a fixed ~0.5 ms pass at C3 that does not shrink with "
                         "the slab: the Amdahl term of this layout); gather_ms / broadcast_x_ms are measured alone, outside the timed region",
                 "broadcast_x_ms": round(bcast_ms, 3), "x_bytes": int(xc.numel() * xc.element_size()),
                 "ms_per_step_incl_replication": round(el / args.steps * 1e3 + bcast_ms, 3),
                 "value_incl_replication": round(I * F / (el / args.steps + bcast_ms * 1e-3) / 1e6, 4)}
    reciprocal = bool(plan.reciprocal)
    folded = bool(plan.folded)
    unfolded_ms = None
    if world == 1 and folded and not args.no_general:          # the same frame with both traces of every pair gathered (rounds 1-3: plan flag QDAS_PLAN_NO_FOLD)
        uplan = DasPlan(prob, device=dev, kernel=args.kernel, reciprocal=True, jit=args.jit, fold=False, **slab_kw)
        uplan.execute_colmajor(xc, F)
        unfolded_ms = kernel_time(uplan, 3)
        uplan.close()
    fold_only_ms = None
    if world == 1 and folded and bool(plan.mirror) and not args.no_general:   # the same frame on the folded kernel WITHOUT the lateral-mirror mode: what a reciprocal acquisition
        fplan = DasPlan(prob, device=dev, kernel=args.kernel, reciprocal=True, jit=args.jit, mirror=False, **slab_kw)   # with a calibrated (not mirror-symmetric) probe costs
        fplan.execute_colmajor(xc, F)
        fold_only_ms = kernel_time(fplan, 3)
        fplan.close()
    general_ms = None
    if world == 1 and reciprocal and not args.no_general:      # the same frame without the reciprocal special case (plan flag)
        gplan = DasPlan(prob, device=dev, kernel=args.kernel, reciprocal=False, jit=args.jit, **slab_kw)       # (no reciprocal mode: no fold either)
        gplan.execute_colmajor(xc, F)
        general_ms = kernel_time(gplan, 3)
        gplan.close()

    # a STREAM of four distinct frames through the same plan (the metric is I * F / t; the reference's own benchmark runs F = 10,
    # test/ParTest.m:244-271): frame pairs share a launch -- tap index and weights once per pair --, so a stream is faster per frame than the
    # single-frame headline above.  Reported beside it, never as `value`; the last frame is compared with its own single-frame execute.
    stream = None
    if world == 1 and F == 1 and not args.no_general and not os.environ.get("QDAS_BENCH_CHILD"):
        try:
            FS = 4
            g2 = torch.Generator(device=dev).manual_seed(4321)
            xs = torch.empty((FS,) + tuple(xc.shape), dtype=xc.dtype, device=dev)
            xs[0].copy_(xc)
            for f in range(1, FS):
                fr = torch.view_as_complex(torch.randn((M, N, T, 2), generator=g2, device=dev, dtype=torch.float32))
                if w["prec"] != "single":
                    from qups_amd.das_spec import _cast_data
                    fr = _cast_data(fr, w["prec"], dev)
                xs[f].copy_(fr.reshape(xc.shape))
            ys = torch.empty((FS, 1, 1, splan.out_count), dtype=xc.dtype, device=dev)
            plan.execute_into(xs, ys, FS)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            reps = 3
            for _ in range(reps):
                plan.execute_into(xs, ys, FS)
            torch.cuda.synchronize()
            sms = (time.perf_counter() - ts) / reps / FS * 1e3
            y1 = torch.view_as_real(plan.execute_colmajor(xs[FS - 1], 1).reshape(-1)).float()
            ya = torch.view_as_real(ys[FS - 1].reshape(-1)).float()
            dev_err = float((ya - y1).abs().max() / y1.abs().max())
            stream = {"frames": FS, "ms_per_frame": round(sms, 3), "value": round(I / (sms * 1e-3) / 1e6, 4), "unit": "Mpixel/s",
                      "last_frame_vs_its_single_execute": float(f"{dev_err:.2e}"),
                      "note": "four distinct frames per call of qdas_plan_execute_frames, wall clock around 3 calls; folds (if any) included"}
            # the same stream through a plan WITHOUT any symmetry mode (no reciprocity fold, no lateral mirror): what an acquisition with a calibrated -- not
            # bit-symmetric -- probe gets.  Four frames share a launch (tap index and weights once per group): the only sharing the general loop has
            general_stream = None
            if (reciprocal_hint := bool(plan.reciprocal) or bool(plan.mirror)):
                try:
                    gs = DasPlan(prob, device=dev, kernel=args.kernel, reciprocal=False, mirror=False, jit=args.jit, **slab_kw)
                    gs.execute_into(xs, ys, FS)
                    torch.cuda.synchronize()
                    ts = time.perf_counter()
                    for _ in range(2):
                        gs.execute_into(xs, ys, FS)
                    torch.cuda.synchronize()
                    gms = (time.perf_counter() - ts) / 2 / FS * 1e3
                    yg1 = torch.view_as_real(gs.execute_colmajor(xs[FS - 1], 1).reshape(-1)).float()
                    general_stream = {"ms_per_frame": round(gms, 3), "frames": FS, "kernel_name": gs.kernel_name(),
                                      "last_frame_vs_its_single_execute": float(f"{float((torch.view_as_real(ys[FS - 1].reshape(-1)).float() - yg1).abs().max() / yg1.abs().max()):.2e}")}
                    gs.close()
                except Exception as ex:
                    general_stream = {"ms_per_frame": None, "note": f"failed: {ex!r}"}
            stream["general_stream"] = general_stream
            del xs, ys
        except Exception as ex:
            stream = {"frames": 4, "ms_per_frame": None, "note": f"failed: {ex!r}"}

    prebuilt_ms = None
    if world == 1 and args.jit and not os.environ.get("QDAS_BENCH_CHILD"):      # the same frame on the prebuilt instantiation
        pplan = DasPlan(prob, device=dev, kernel=args.kernel, reciprocal=not args.no_reciprocal, jit=False, fold=not args.no_fold, **slab_kw)
        pplan.execute_colmajor(xc, F)
        prebuilt_ms = kernel_time(pplan, 3)
        pplan.close()

    traffic, tsrc, ctrs = None, "not measured", {}
    if rank == 0 and not os.environ.get("QDAS_BENCH_CHILD"):
        mode = args.traffic
        if mode == "auto":
            mode = "live" if world == 1 else "file"
        if mode == "live":
            argv = ["--workload", args.workload] + (["--kernel", str(args.kernel)] if args.kernel else []) + \
                   (["--prec", args.prec] if args.prec else []) + (["--fmod", str(args.fmod)] if args.fmod else []) + (["--rx-apod", args.rx_apod] if args.rx_apod else []) + (["--rx-apod-array"] if args.rx_apod_array else []) + (["--window-apod"] if args.window_apod else []) + (["--tx-apod", args.tx_apod] if args.tx_apod else []) + (["--gen-apod"] if args.gen_apod else []) + \
                   (["--no-reciprocal"] if args.no_reciprocal else []) + (["--no-jit"] if args.no_jit else []) + (["--no-fold"] if args.no_fold else []) + \
                   (["--frames", str(F)] if F > 1 else [])
            traffic, tsrc, ctrs = measure_traffic(argv)
            if traffic is None:
                mode = "file"
                tsrc += "; "
            else:
                tsrc = tsrc
        if mode == "file":
            tfile = os.path.join(ROOT, "profiles", f"traffic_{w['name']}.json")
            if os.path.exists(tfile):
                try:
                    traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
                    tsrc = (tsrc if tsrc.endswith("; ") else "") + f"profiles/traffic_{w['name']}.json (committed rocprofv3 pass, not this run)"
                except Exception:
                    traffic = None

    if rank == 0:
        ms = el / args.steps * 1e3
        I_step = I * F                                             # pixels beamformed per step
        pairs = I * N * M
        sb = {"halfT": 4, "single": 8, "double": 16}[w["prec"]]
        apb = 0 if w["apod"] is None else w["apod"].size * sb // 2
        alg_bytes = (T * N * M * sb + 12 * I + sb * I + apb) / world      # per launch (per rank): x + Pi + y (+ apod)  (SURVEY 8d "B")
        info = _lib.device_info(local)
        ksec = kernel_ms * 1e-3
        mirror = bool(plan.mirror)
        exec_fpp = exec_flop_per_pair(w["interp"], (2 if mirror else 1) if folded else (2 if reciprocal else 1) * (2 if mirror else 1), folded)
        # pairs the kernel really executes: a pixel x receiver weight (array or generated rule) drops whole (wave, receiver) stages
        exec_frac, mask = 1.0, None
        try:
            if args.rx_apod and not args.rx_apod_array and plan.kernel == "tiled":
                from qups_amd import apodization as A
                kind, _, par = args.rx_apod.partition(":")
                fn = {"acceptance": A.ap_acceptance_angle, "cosine": A.ap_cosine_angle, "fnumber": A.ap_aperture_growth}[kind]
                nrm = w.get("nrm")
                if nrm is None:
                    nrm = np.tile(np.array([[0.0], [0.0], [1.0]]), (1, N))
                mask = fn(w["Pi"], w["Pr"], nrm, *([float(par)] if par else []))
            elif plan.kernel == "tiled" and not args.tx_apod:
                for a in opts["apod"]:
                    a = np.asarray(a)
                    if a.ndim >= 4 and a.shape[0] == w["I1"] and a.shape[1] == w["I2"] and a.shape[3] == N and (a.ndim < 5 or a.shape[4] == 1):
                        mask = a if mask is None else mask * a
            if mask is not None:
                exec_frac = executed_pair_fraction(np.asarray(mask).reshape(w["I1"], w["I2"], N), plan.wave_shape())
        except Exception:
            exec_frac = None
        rec = {
            "metric": "beamformed Mpixels/sec (1024^2 px, 256x256 Tx/Rx)" if w["name"] == "c3" else "beamformed Mpixels/sec",
            "value": round(I_step / (el / args.steps) / 1e6, 4), "unit": "Mpixel/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": {"halfT": "f16", "single": "f32", "double": "f64"}[w["prec"]], "data": "synthetic",
            "config": {"workload": w["label"], "pixels": I, "pairs_per_frame": pairs, "kernel": plan.kernel, "reciprocal_mode": reciprocal, "reciprocity_fold": folded, "mirror_mode": mirror,
                       "kernel_name": plan.kernel_name(), "jit": bool(args.jit),
                       "fallback_tiles": fallback, "tile": list(plan.tile_shape()), "wave": list(plan.wave_shape()), "aperture_split": plan.aperture_split(), "parallelism": (f"{'mirror-' if splan.mirror_slabs else ''}pixel-slab x{world} + RCCL all_gather") if world > 1 else "1 GPU",
                       "device": info["name"], "cu": info["cu_count"]},
            "roofline": {"bound": "hbm", "achieved": round(alg_bytes / ksec / 1e9, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(alg_bytes / ksec / 1e9 / HBM_PEAK_GBS, 6),
                         "traffic": traffic, "traffic_source": tsrc, "kernel_ms": round(kernel_ms, 3), "algorithmic_bytes": int(alg_bytes),
                         **({} if not apb else {"algorithmic_bytes_reference_convention": int(alg_bytes + apb / world),
                                                "algorithmic_bytes_note": "algorithmic_bytes counts the apodization array as this ABI takes it -- REAL weights in the data precision (apod_real), "
                                                                          "half the bytes; the reference casts weights to complex (kern/das_spec.m:243,345), which is what SURVEY 8d / BASELINE.md price "
                                                                          "(algorithmic_bytes_reference_convention)"}),
                         "note": "compulsory-traffic accounting: this path is FP32-VALU / LDS-gather bound "
                                 "(~2.5e3 flop/byte): valu_frac_executed = flops the pair loop executes / fp32 vector peak; "
                                 "reference_model_tflops = the REFERENCE kernel's per-pair flop count (SURVEY 8d) x pairs / time -- an equivalent rate that can exceed "
                                 "the vector peak because reciprocal / mirror modes share index and weight work between traces",
                         "gpairs_per_s": round(pairs / world / ksec / 1e9, 3),
                         "reference_model_tflops": round(pairs / world * FLOP_PER_PAIR.get(w["interp"], 42) / ksec / 1e12, 3),
                         "valu_flop_per_pair_executed": exec_fpp,
                         "pairs_executed_frac": None if exec_frac is None else round(exec_frac, 4),
                         "valu_frac_executed": None if exec_fpp is None or exec_frac is None or plan.kernel != "tiled" or args.tx_apod else
                                               round(pairs / world * exec_frac * exec_fpp / ksec / 1e12 / FP32_PEAK_TFLOPS, 4)},
        }
        try:
            taps = {"nearest": 1, "linear": 2, "cubic": 4, "lanczos3": 4}.get(w["interp"], 4)
            rec["roofline"].update(binding_roofs(ctrs, info["cu_count"], pairs / world * (exec_frac or 1.0) * ((N + 1) / (2.0 * N) if folded else 1.0), taps, sb))
        except Exception as ex:
            rec["roofline"]["binding_note"] = f"counter post-processing failed: {ex!r}"
        # The roof that BINDS: VALU issue under the part's power limit.  achieved = wave64 VALU instructions the dominant kernel issued
        # (SQ_INSTS_VALU of the counter pass of this command) / its launch time; peak = the SATURATED issue rate of the pair loop's own VALU mix, measured on
        # this device in this run (csrc/probe.hip qdas_debug_issue_rate, mix 2: 16 broadcast pk_fma + 16 SGPR-coefficient pk_fma + 5 index instructions, four
        # waves per SIMD on every CU, no LDS reads -- so frac < 1 also prices the gathers, the staging and the per-stage code).  The BASELINE's HBM roof stays
        # in the line as nominal_hbm_*: SURVEY 8d predicted, and every round measured, that compulsory traffic is ~1 % of it.
        rf = rec["roofline"]
        rf["nominal_hbm_achieved"], rf["nominal_hbm_peak"], rf["nominal_hbm_frac"], rf["nominal_hbm_unit"] = rf["achieved"], rf["peak"], rf["frac"], "GB/s"
        # (not in the counter passes' child runs: the probe's launches would be the longest kernels of a short workload's trace)
        sat = None if os.environ.get("QDAS_BENCH_CHILD") else saturated_issue_rate(_lib.lib(), info["cu_count"])
        if sat and plan.kernel == "tiled" and "valu_insts" in rf and rf.get("counter_pass_kernel_ms"):
            ach = rf["valu_insts"] / (rf["counter_pass_kernel_ms"] * 1e-3) / 1e9
            rf.update({"bound": "valu_issue", "achieved": round(ach, 2), "peak": round(sat["pair_mix_ginst_s"], 2), "unit": "Gwave-inst/s", "frac": round(ach / sat["pair_mix_ginst_s"], 4),
                       "peak_source": sat["note"], "saturated_ns_per_inst_simd": sat["ns"]})
            rf.pop("valu_issue_frac", None)             # (rounds 3-4: instructions x 4 cycles / active cycles -- a convention, not a measured roof)
            rf["binding_note"] = ("VALU issue under the power limit binds (frac: against the measured saturated rate of the loop's own VALU mix on this box); LDS gathers second "
                                  "(lds_busy_frac); nominal_hbm_frac is the BASELINE's roof, unreachable at ~2.5e3 flop/byte")
        elif sat:
            rf["bound_note"] = "bound = hbm is the BASELINE's nominal roof only: no SQ counters in this run to price the VALU-issue roof that binds"
            rf["saturated_ns_per_inst_simd"] = sat["ns"]
        if F > 1:
            rec["frames_per_step"] = F
            rec["ms_per_frame"] = round(ms / F, 3)
            rec["config"]["workload"] += f" [stream of {F} distinct frames per step]"
            rec["roofline"]["kernel_ms_is"] = "per frame (kernel time of the step / frames)"
        if stream is not None:
            rec["stream"] = stream
        if prebuilt_ms is not None:
            rec["prebuilt_kernel_ms"] = round(prebuilt_ms, 3)
        if folded:
            rec["roofline"]["fold_note"] = ("reciprocal acquisition: every step first adds the two traces of each unordered transmit / receive pair (fold.hip: one pass over "
                                            "HBM, inside the timed region and inside kernel_ms), then the fused kernel sums the N(N+1)/2 folded traces -- the image is the "
                                            "full N x M sum (linearity of the interpolators; parity_check on random, non-symmetric data); products_formed_frac = "
                                            "share of the N x M products formed per pixel; unfolded_ms_per_step = the same frame with the plan flag QDAS_PLAN_NO_FOLD")
            rec["roofline"]["products_formed_frac"] = round((N + 1) / (2.0 * N), 4)
        if unfolded_ms is not None:
            rec["unfolded_ms_per_step"] = round(unfolded_ms, 3)
        if fold_only_ms is not None:
            rec["fold_only_ms_per_step"] = round(fold_only_ms, 3)
        if general_ms is not None:
            rec["general_ms_per_step"] = round(general_ms, 3)
            rec["general_value"] = round(I / (general_ms * 1e-3) / 1e6, 4)
        if stream and stream.get("general_stream") and stream["general_stream"].get("ms_per_frame") is not None:
            rec["general_stream_ms_per_frame"] = stream["general_stream"]["ms_per_frame"]      # (no symmetry mode at all, four frames per launch: beside the headline, never `value`)
        if multi:
            rec["multi_gpu"] = multi
        if world == 1 and not args.no_cpu:
            try:
                xlast = xc if F == 1 else xc[F - 1]
                xh = torch.view_as_real(xlast).float().cpu().numpy().view(np.complex64).reshape(M, N, T).transpose(2, 1, 0)
                plain = not (args.fmod or args.rx_apod or args.window_apod or args.tx_apod or args.gen_apod)
                cb, lstep, limg = cpu_baseline(w, xh, apod=w["apod"] if plain else None)
                rec["cpu_baseline"] = cb[0]
                if len(cb) > 1:
                    rec["cpu_baseline_tuned"] = cb[1]
                # parity of the very frame that was timed (last timed step, the kernel named in config.kernel_name) against the CPU
                # port on the baseline's pixel lattice -- outside the timed region
                if plain and limg is not None:
                    from oracle import das_ref
                    img = torch.view_as_real(yimg.reshape(F, -1)[F - 1]).float().cpu().numpy().view(np.complex64).reshape(w["I1"], w["I2"], order="F")
                    den = float(np.abs(limg).max())
                    err32 = float(np.abs(img[::lstep, ::lstep] - limg).max()) / (den if den > 0 else 1.0)
                    # the judge of parity is the DOUBLE-precision port (the float32 port's own rounding -- fp32 delays at tau*fs ~ 3e3
                    # samples -- is ~1.5e-4 of the maximum at C3, above the kernel's): every other lattice pixel per axis
                    s2 = 2 * lstep
                    ap = () if w["apod"] is None else (w["apod"][::s2, ::s2],)
                    ref = das_ref.das_spec("DAS", w["Pi"][:, ::s2, ::s2, :], w["Pr"], w["Pv"], w["Nv"], xh, w["t0"], w["fs"],
                                           1.0 / np.float64(np.float32(1.0 / w["c0"])), VS="plane-waves" not in w["opt"],
                                           DV="diverging-waves" in w["opt"], interp=w["interp"], apod=ap, prec="double")[:, :, 0, 0, 0]
                    den64 = float(np.abs(ref).max())
                    err = float(np.abs(img[::s2, ::s2] - ref).max()) / (den64 if den64 > 0 else 1.0)
                    tol = 2e-3 if w["prec"] == "halfT" else 1e-4
                    rec["parity_check"] = {"rel_err": float(f"{err:.3e}"), "tol": tol, "pixels": int(ref.size), "ok": bool(err <= tol and den64 > 0),
                                           "against": "oracle/das_ref.c in double precision (the C restatement the cpu_baseline times in float32) on "
                                                      "every %dth pixel per axis of the frame of the last timed step; max |gpu - cpu| / max |cpu|" % s2,
                                           "rel_err_vs_float32_port": float(f"{err32:.3e}"), "pixels_float32_port": int(limg.size),
                                           "kernel": plan.kernel_name()}
            except Exception as ex:  # report, never hide
                rec["cpu_baseline"] = {"value": None, "unit": "Mpixel/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"FAILED: {ex!r}"}
        if args.checksum:
            raw = torch.view_as_real(yimg.reshape(-1)).cpu().numpy().tobytes()
            try:
                import xxhash
                rec["image_checksum"] = xxhash.xxh3_128_hexdigest(raw)
            except ImportError:
                import hashlib
                rec["image_checksum"] = hashlib.blake2b(raw, digest_size=16).hexdigest()
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
