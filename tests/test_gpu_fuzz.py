"""Randomised differential test (GPU): the fused tiled kernel against the float64 oracle over random shapes,
sequences, interpolators, layouts, weights, shards and forced kernel shapes.  Every case is seeded; a failure
prints the seed and the drawn configuration."""
from __future__ import annotations

import os

import numpy as np
import pytest

from tests.cases import cinv_f32, make_case, rel_err

pytestmark = pytest.mark.gpu


def _draw(seed):
    r = np.random.default_rng(1000 + seed)
    seq = r.choice(["FSA", "PW", "DV", "FC"])
    interp = r.choice(["nearest", "linear", "cubic", "lanczos3", "cubic_dev"])
    N = int(r.choice([1, 2, 3, 7, 16, 17, 32, 33, 48]))
    M = int(r.choice([1, 2, 5, 16, 31, 32, 33, 40]))
    I1 = int(r.integers(1, 200))
    I2 = int(r.integers(1, 40))
    cfg = dict(seq=seq, interp=interp, N=N, M=M, I1=I1, I2=I2,
               tpose=bool(r.integers(0, 2)), fmod=float(np.float32(r.choice([0.0, 0.0, 2.5e6]))),
               prec=str(r.choice(["single", "single", "halfT"])), data=str(r.choice(["targets", "noise"])),
               t0vec=bool(r.integers(0, 3) == 0), wn=bool(r.integers(0, 3) == 0), wm=bool(r.integers(0, 3) == 0),
               wpix=bool(r.integers(0, 3) == 0), shard=bool(r.integers(0, 3) == 0),
               tz=int(r.choice([0, 0, 64, 32, 16, 8])), ks=int(r.choice([0, 0, 1, 2, 3, 4])),
               fun=str(r.choice(["DAS", "DAS", "SYN", "MUL"])), F=int(r.choice([1, 1, 2, 3, 4, 6])), cmap=bool(r.integers(0, 4) == 0),
               gen=str(r.choice(["", "", "", "acceptance", "cosine", "fnumber"])))
    # (round 2; drawn last so that the fields above keep their round-1 values per seed)
    extra = dict(sym=bool(r.integers(0, 6) == 0), bf=bool(r.integers(0, 8) == 0), jit=bool(r.integers(0, 10) == 0))
    if extra["sym"]:                # a reciprocal-mode candidate: full synthetic aperture, M == N, no pixel-dependent weights, plain 'DAS'
        cfg.update(seq="FSA", M=N, wpix=False, gen="", fun="DAS")       # (pixel-independent weights wn / wm stay: reciprocal mode with a weight table)
    elif extra["bf"] and cfg["prec"] == "single":
        cfg.update(fun="BF", F=min(cfg["F"], 2))
    cfg.update(extra)
    if r.integers(0, 8) == 0 and not extra["sym"] and cfg["fun"] != "BF":   # fp64 data on the fused kernel: 'DAS', pixel-independent weights,
        cfg.update(prec="double", wpix=cfg["wpix"] and cfg["fmod"] == 0.0, gen="", fun="DAS", cmap=False, jit=False)   # remodulation OR one pixel x receiver array (round 3: the drawn fmod / wpix stay)
    # a pixel x TRANSMIT weight (scanline-style transmit apodization): fused with the roles of the apertures swapped
    cfg["wpm"] = bool(r.integers(0, 10) == 0) and cfg["fun"] == "DAS" and not cfg["wpix"] and not cfg["gen"] and not cfg["sym"] and cfg["prec"] != "double"
    # pixel pitch: ~lambda/3 (2.6 samples of delay per pixel), ~lambda (the 384-sample windows of the second attempt), ~1.6 lambda (tiles that fall back)
    cfg["coarse"] = int(r.choice([1, 1, 1, 1, 3, 5]))
    cfg["fold"] = bool(r.integers(0, 3) == 0)        # a second pixel-dependent array on the same side (per depth x element): folded per plan
    if cfg["prec"] == "double":
        cfg["fold"] = False                           # (fp64 data: one pixel x receiver array, used in place)
    # a transmit-side AND a receive-side pixel array (real weights, fp32 data, plain 'DAS'): per-pair pixel weights on the wide-window configuration
    if r.integers(0, 12) == 0 and cfg["prec"] == "single" and not cfg["sym"] and not cfg["bf"] and cfg["N"] > 1 and cfg["M"] > 1:
        cfg.update(wpix=True, wpm=True, wm=False, gen="", fun="DAS", fmod=0.0)
    # (round 3; its own generator, so that every field above and the data drawn from `r` keep their values per seed)  The HEADLINE kernel
    # variant: reciprocal mode AND hiprtc build AND N % 32 == 0 -> 32-transmit stages, a configuration that exists only as a hiprtc build
    # (what bench.py times at C3).  ~1 in 10 seeds; the default 128 seeds hold more than ten of them.
    r3 = np.random.default_rng(77000 + seed)
    cfg["headline"] = bool(r3.integers(0, 10) == 0) and cfg["prec"] != "double"
    if cfg["headline"]:
        n3 = int(r3.choice([32, 32, 64]))
        cfg.update(seq="FSA", N=n3, M=n3, sym=True, jit=True, bf=False, wpix=False, wpm=False, gen="", fun="DAS", cmap=False, coarse=1,
                   F=min(cfg["F"], 2 if n3 == 64 else 3), I1=max(cfg["I1"], 40), tz=0)
    if os.environ.get("QDAS_FUZZ_OVERRIDE"):                            # debugging aid: JSON dict of fields to force
        import json
        cfg.update(json.loads(os.environ["QDAS_FUZZ_OVERRIDE"]))
    return r, cfg


# Seeds (and frames) that needed the shifted-oracle explanation below -- the one place where an edge-rule bug could hide: counted,
# reported and bounded by test_fuzz_escape_rate at the end of this file.
_ESCAPES: list = []
_RAN: list = []
ESCAPE_RATE_MAX = 0.05          # of the seeds that ran; rounds 1-3 saw well under 1 % in soaks of > 100 000 seeds

_SEED0 = int(os.environ.get("QDAS_FUZZ_OFFSET", "0"))          # soak runs: QDAS_FUZZ_OFFSET=2000 QDAS_FUZZ_SEEDS=2000 pytest -n 8 ...


# Seeds that found a bug once, run with every suite whatever the offset.  50160 (round 6): 64-transmit stages of a one-set hiprtc build with a
# transmit weight table holding a zero and a split aperture -- the stage's non-zero masks were 32 bits wide and the image came out all zero.
_PINNED = [50160]          # (126301 with hiprtc builds forced -- roles swapped onto ONE transmit, images that differed from run to run: tests/test_gpu_edges.py)
_SEEDS = list(range(_SEED0, _SEED0 + int(os.environ.get("QDAS_FUZZ_SEEDS", "128"))))


@pytest.mark.parametrize("seed", _SEEDS + [s for s in _PINNED if s not in _SEEDS])
def test_tiled_kernel_random_configuration(seed, monkeypatch):
    import torch
    from qups_amd import DasPlan, build_problem, parse_options
    from qups_amd.das_spec import _cast_data, _colmajor
    from oracle import das_oracle as O
    r, c = _draw(seed)
    _RAN.append(seed)
    zspan = (4e-3, 4e-3 + max(c["I1"], 2) * 0.1e-3 * c["coarse"])
    case = make_case(seq=c["seq"], interp=c["interp"], seed=seed, N=c["N"], M=c["M"], I1=c["I1"], I2=c["I2"], zlim=zspan,
                     xspan=2e-3 * c["coarse"], data=c["data"])
    N, M = case["N"], case["M"]
    x = case["x"]
    if c["prec"] == "halfT":
        x = (x.real.astype(np.float16).astype(np.float32) + 1j * x.imag.astype(np.float16).astype(np.float32)).astype(np.complex64)
    t0 = case["t0"]
    if c["t0vec"]:
        t0 = (case["t0"] + np.float32(1.0 / case["fs"]) * r.integers(-3, 4, (1, 1, M))).astype(np.float32).astype(np.float64)
    apod = []
    q = (lambda a: a.astype(np.float16).astype(np.float64)) if c["prec"] == "halfT" else (lambda a: a.astype(np.float32).astype(np.float64))
    if c["prec"] == "double":
        q = lambda a: a.astype(np.float64)
    if c["wn"]:
        apod.append(q(r.uniform(0.2, 1, (1, 1, 1, N, 1))))
    if c["wm"]:
        a = q(r.uniform(0, 1, (1, 1, 1, 1, M)))
        a[..., r.integers(0, M)] = 0.0
        apod.append(a * (1 + 0.5j) if c["prec"] == "single" else a)
    if c["wpix"] and N > 1:
        a = q(r.uniform(0, 1, (c["I1"], c["I2"], 1, N, 1)) > 0.4)
        a[: c["I1"] // 3] = 0.0
        apod.append(a)
    if c["wpm"] and M > 1:
        a = q(r.uniform(0, 1, (c["I1"], c["I2"], 1, 1, M)) > 0.5)
        a[c["I1"] // 2:, :, :, :, 0] = 0.0
        apod.append(a)
        if c["fold"]:
            apod.append(q(r.uniform(0.5, 1, (c["I1"], 1, 1, 1, M))))
    if c["wpix"] and N > 1 and c["fold"]:
        apod.append(q(r.uniform(0.5, 1, (c["I1"], 1, 1, N, 1))))
    fun = c["fun"] if c["prec"] == "single" else "DAS"                # 'SYN' / 'MUL' are fused for fp32 data only
    if fun in ("MUL", "BF") and ((c["wpix"] and N > 1) or c["gen"]):
        fun = "SYN"                                                   # 'MUL' with pixel x receiver weights: generic kernel
    F = c["F"]
    xs_all = [x] + [(r.standard_normal(x.shape) + 1j * r.standard_normal(x.shape)).astype(np.complex64) for _ in range(F - 1)]
    if c["prec"] == "halfT":
        xs_all = [(v.real.astype(np.float16).astype(np.float32) + 1j * v.imag.astype(np.float16).astype(np.float32)).astype(np.complex64) for v in xs_all]
    cval = case["c"]
    c_or = cinv_f32(case["c"]) if c["prec"] != "double" else case["c"]
    if c["cmap"]:                                                        # smooth per-pixel sound-speed map
        zz, xx = np.meshgrid(np.linspace(0, 1, c["I1"]), np.linspace(0, 1, c["I2"]), indexing="ij")
        f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
        cval = f32(1.0 / f32(1.0 / (1540.0 + 20.0 * np.sin(1.9 * zz + 0.3) * np.cos(1.3 * xx))))[:, :, None]
        c_or = 1.0 / f32(1.0 / cval)
    gen_spec = None
    if c["gen"] and not (c["wpix"] and N > 1):
        from qups_amd import apodization as A
        from qups_amd import geometry as G
        nrm = np.asarray(G.linear_array(N, 0.3e-3)[1], np.float32).astype(np.float64)
        kw = dict(theta=35.0) if c["gen"] != "fnumber" else dict(f=1.0, Dmax=5e-3)
        gen_spec = A.rx_apod_spec(c["gen"], normals=nrm, **kw)
        gen_arr = {"acceptance": lambda: A.ap_acceptance_angle(case["Pi"], case["Pr"], nrm, 35.0),
                   "cosine": lambda: A.ap_cosine_angle(case["Pi"], case["Pr"], nrm, 35.0),
                   "fnumber": lambda: A.ap_aperture_growth(case["Pi"], case["Pr"], nrm, 1.0, 5e-3)}[c["gen"]]()
    if c["tz"]:
        monkeypatch.setenv("QDAS_TILE_Z", str(c["tz"]))
    if c["ks"]:
        monkeypatch.setenv("QDAS_KSPLIT", str(c["ks"]))
    xs = np.stack([np.swapaxes(v, 1, 2) if c["tpose"] else v for v in xs_all], axis=3)
    opts = list(case["opt"]) + ["interp", c["interp"], "input-precision", c["prec"], "modulation", c["fmod"], "transpose", c["tpose"]]
    for a in apod:
        opts += ["apod", a]
    if gen_spec is not None:
        opts += ["rx-apod", gen_spec]
    I = c["I1"] * c["I2"]
    kw = {}
    if c["shard"] and I >= 3:
        kw = dict(i_begin=I // 3, i_count=I - I // 3 - I // 4)
    xt = torch.from_numpy(np.ascontiguousarray(xs))
    po = parse_options(xt, opts)
    prob = build_problem(fun, case["Pi"], case["Pr"], case["Pv"], case["Nv"], tuple(xt.shape), t0, case["fs"], cval, po)
    plan = DasPlan(prob, kernel=2, jit=c["jit"], **kw)
    if os.environ.get("QDAS_FUZZ_DEBUG"):
        print("plan:", plan.kernel_name(), "tile", plan.tile_shape(), "wave", plan.wave_shape(), "split", plan.aperture_split(), "fallback tiles", plan.fallback_tiles())
    if c["jit"] and fun != "BF":                                        # ('BF' runs the prebuilt kernel; so does a plan whose hiprtc build would spill registers)
        assert "[jit " in plan.kernel_name() or "scratch" in plan.jit_note(), (c, plan.kernel_name(), plan.jit_note())
    if c.get("headline") and not c["t0vec"]:                            # (a per-transmit t0 is not reciprocal: general kernel)
        # (mirror-symmetric draws -- no weights, one t0, the whole image -- run reciprocal + lateral-mirror mode: four sets of 16 transmits)
        # (fp32: the reciprocity-folded frame -- 32-transmit stages with or without the mirror mode (16 when its tiles need 192-sample windows);
        #  fp16 data keep both traces: four window sets of 16 transmits in mirror mode, 32-transmit stages otherwise)
        if c["prec"] == "single":
            assert plan.reciprocal and plan.folded and ",fold" in plan.kernel_name(), (c, plan.kernel_name())
        else:           # (fp16 data: folded into complex64, the folded fp32 kernels)
            assert plan.reciprocal and plan.folded and "f16>f32" in plan.kernel_name(), (c, plan.kernel_name())
    xc = _colmajor(_cast_data(xt, prob.prec, plan.device))
    y = plan.execute_colmajor(xc, F)                                    # (F, oM, oN, count)
    torch.cuda.synchronize()
    outs = y.to(torch.complex128 if c["prec"] == "double" else torch.complex64).cpu().numpy()
    assert plan.kernel == "tiled", c
    oapod = tuple(apod) + ((gen_arr,) if gen_spec is not None else ())
    for f in range(F):
        ref = O.das_spec(fun, case["Pi"], case["Pr"], case["Pv"], case["Nv"], xs_all[f], t0, case["fs"], c_or,
                         VS=case["VS"], DV=case["DV"], interp=c["interp"], apod=oapod, fmod=c["fmod"])
        if fun == "BF" and c["tpose"]:
            ref = np.swapaxes(ref, -1, -2)                                # planes in the data's aperture order (src/bf.cu:99,135)
        refv = ref.reshape(I, -1, order="F")                              # pixels x planes
        if kw:
            refv = refv[kw["i_begin"]: kw["i_begin"] + kw["i_count"]]
        out = outs[f].reshape(refv.shape[1], -1).T                        # (planes, count) -> count x planes
        den = np.abs(ref).max()
        if den == 0:
            assert np.abs(out).max() == 0, c
            continue
        err = np.abs(out - refv).max() / den
        if c["interp"] == "nearest" or (gen_spec is not None and c["gen"] != "cosine"):
            # discontinuous in tau (nearest) or in the geometry (binary masks): a rounding at the step swaps ONE of the N*M
            # samples of a pixel / one receiver; such pixels are rare, the others agree to rounding
            bad = np.abs(out - refv) / den > {"halfT": 3e-3, "single": 1e-4, "double": 1e-9}[c["prec"]]
            assert bad.mean() <= 0.05 * c["coarse"], (seed, c, float(bad.mean()))     # (longer records: fp32 delays round more often across a step)
            continue
        tol = 3e-3 if c["prec"] == "halfT" else 1e-4                      # covers tiles that fell back to the generic kernel (fp32 delays)
        if c["prec"] == "double":
            tol = 1e-9
        elif plan.fallback_tiles():
            tol *= (1 if c["coarse"] == 1 else 2 * c["coarse"]) * (2.5 if c["fmod"] else 1.0)              # fp32 delays (and the modulation phase 2 pi fmod tau formed from them) round in
                                                                          # proportion to the record length, which grows with the pitch
        if err > tol and os.environ.get("QDAS_FUZZ_DEBUG"):                 # debugging aid: where, and what the generic kernel says
            yg = DasPlan(prob, kernel=1, **kw).execute_colmajor(xc, F).to(torch.complex64).cpu().numpy()[f].reshape(refv.shape[1], -1).T
            e = np.abs(out - refv).max(axis=1) / den
            worst = np.argsort(e)[::-1][:12]
            i0 = kw.get("i_begin", 0)
            print("bad pixels:", int((e > tol).sum()), "of", e.size, "frame", f, "|", plan.kernel_name(), "tile", plan.tile_shape(), "wave", plan.wave_shape(),
                  "split", plan.aperture_split(), "fallback tiles", plan.fallback_tiles())
            for w in worst:
                print("  i1", int((w + i0) % c["I1"]), "col", int((w + i0) // c["I1"]), "err", float(e[w]), "tiled", out[w, 0], "generic", yg[w, 0], "oracle", refv[w, 0])
        if err > tol:
            # The edge rule (all taps inside the record, else exactly 0) is a step in tau: a pair whose first / last tap sits within
            # rounding of the end of the record is counted by one precision and not by the other.  Such pixels must be rare and
            # must be explained by the float64 oracle with its time origin moved by 2e-4 samples either way.
            e = np.abs(out - refv).max(axis=1) / den
            bad = np.nonzero(e > tol)[0]
            _ESCAPES.append(dict(seed=int(seed), frame=int(f), pixels=int(bad.size), of=int(e.size), err=float(err), tol=float(tol), seq=str(c["seq"]), interp=str(c["interp"])))
            # (focused waves: the delay changes sign with (Pi - Pv).Nv, src/bf.cu:107 -- at the focal depth the fp32 geometry of the kernels and the
            #  float64 oracle may disagree on the sign of a dot product that is ~0: a row of pixels, visible pair by pair in 'BF')
            lim = max(2, e.size // 500) if c["seq"] != "FC" else max(4, e.size // 150)
            assert bad.size <= lim, (plan.kernel_name(), seed, c, f, plan.tile_shape(), plan.wave_shape(), plan.aperture_split(), plan.fallback_tiles(), err, bad.size)
            best = e[bad]
            for sh in (-2e-4, 2e-4):
                r2 = O.das_spec(fun, case["Pi"], case["Pr"], case["Pv"], case["Nv"], xs_all[f], np.asarray(t0, np.float64) + sh / case["fs"], case["fs"], c_or,
                                VS=case["VS"], DV=case["DV"], interp=c["interp"], apod=oapod, fmod=c["fmod"]).reshape(I, -1, order="F")
                if kw:
                    r2 = r2[kw["i_begin"]: kw["i_begin"] + kw["i_count"]]
                best = np.minimum(best, np.abs(out[bad] - r2[bad]).max(axis=1) / den)
            assert best.max() <= 10 * tol, (seed, c, f, plan.tile_shape(), plan.wave_shape(), plan.aperture_split(), plan.fallback_tiles(), err, float(best.max()))


def test_fuzz_escape_rate():
    """How often did the shifted-oracle explanation fire?  (it is the only place a real edge-rule bug could hide.)  Prints the
    census, leaves it in ``gpurun_out/fuzz_escapes.json`` when that directory exists, and fails above ESCAPE_RATE_MAX of the seeds."""
    import json
    import warnings
    seeds = sorted({e["seed"] for e in _ESCAPES})
    rec = dict(seeds_run=len(_RAN), seeds_with_escape=len(seeds), events=_ESCAPES, rate_max=ESCAPE_RATE_MAX)
    msg = f"fuzz: {len(seeds)} of {len(_RAN)} seeds took the shifted-oracle escape ({len(_ESCAPES)} frames): seeds {seeds}"
    print(msg)
    warnings.warn(msg)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "fuzz_escapes.json"), "w") as fh:
            json.dump(rec, fh, indent=1)
    if _RAN:
        assert len(seeds) <= max(1, int(ESCAPE_RATE_MAX * len(_RAN))), msg
