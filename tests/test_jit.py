"""QDAS_PLAN_JIT: the tiled kernel compiled by hiprtc with the plan's sizes as constants -- the reference's const-compile
specialisation (src/UltrasoundSystem.m:5626-5748 getDASConstCudaDef, src/sizes.cu:17-52; its own check: test/ParTest.m:322-327)."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.cases import cinv_f32, make_case, rel_err


def test_hiprtc_builds_the_specialised_kernel_without_a_device(tmp_path, monkeypatch):
    """(CPU) libqdas.so carries the kernel headers and hiprtc compiles a specialisation of them for gfx950 -- no spills, no scratch"""
    import sys
    from qups_amd import _lib
    monkeypatch.setenv("QDAS_CACHE_DIR", str(tmp_path))
    L = _lib.lib()
    f = L.qdas_debug_jit_compile
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_ulonglong, C.c_ulonglong, C.c_ulonglong, C.c_char_p, C.c_size_t, C.POINTER(C.c_ulonglong)]
    monkeypatch.setenv("QDAS_JIT_NARROW", "1")            # the reciprocal build plans use: 128-sample windows, 32 transmits per stage
    monkeypatch.setenv("QDAS_JIT_MB", "32")
    for interp, dtype, sym, fmod in ((3, 1, 1, 0), (2, 2, 0, 1)):
        msg, n = C.create_string_buffer(4000), C.c_ulonglong()
        rc = f(interp, dtype, sym, fmod, 64, 64, 1024, msg, 4000, C.byref(n))
        if rc and b"hiprtc not available" in msg.value:
            pytest.skip(msg.value.decode())
        assert rc == 0, msg.value.decode()
        assert n.value > 10000 and os.path.exists(tmp_path / (msg.value.decode() + ".hsaco"))     # cached on disk under its key
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import kernel_regs
    rows = [r for fn in os.listdir(tmp_path) for r in kernel_regs.kernel_table(str(tmp_path / fn))]
    assert len(rows) == 2 and all(r["name"] == "qdas_jit_tile" and r["vgpr_spill"] == 0 and r["scratch"] == 0 for r in rows), rows


@pytest.mark.gpu
@pytest.mark.parametrize("seq,interp,prec,extra", [
    ("FSA", "lanczos3", "single", {}), ("PW", "cubic", "single", {}), ("FC", "linear", "single", {}), ("DV", "nearest", "single", {}),
    ("FSA", "cubic", "halfT", {}), ("PW", "lanczos3", "single", {"fmod": 2.5e6}), ("FSA", "linear", "single", {"noreci": True}),
])
def test_jit_plan_matches_prebuilt_and_oracle(seq, interp, prec, extra, tmp_path, monkeypatch):
    import torch
    from oracle import das_oracle as O
    from qups_amd import DasPlan, build_problem, parse_options
    monkeypatch.setenv("QDAS_CACHE_DIR", str(tmp_path))
    case = make_case(seq=seq, interp=interp, seed=31, N=16, I1=150, I2=37)
    x = torch.from_numpy(case["x"])
    fmod = float(np.float32(extra.get("fmod", 0.0)))
    va = list(case["opt"]) + ["interp", interp, "input-precision", prec, "modulation", fmod]
    opts = parse_options(x, va)
    T, N, M = case["x"].shape
    prob = build_problem("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], (T, N, M), case["t0"], case["fs"], case["c"], opts)
    ys, names = [], []
    for jit in (False, True):
        plan = DasPlan(prob, kernel=2, jit=jit, reciprocal=not extra.get("noreci", False))
        y = plan.feval(x)
        ys.append(torch.view_as_real(y).float().cpu().numpy().view(np.complex64)[..., 0] if prec == "halfT" else y.cpu().numpy())
        names.append(plan.kernel_name())
        plan.close()
    assert "[prebuilt]" in names[0] and "[jit " in names[1], names          # the hiprtc kernel really ran
    assert len(os.listdir(tmp_path)) == 1                                    # ... and its code object is in the disk cache
    ref = O.das_spec("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], case["x"], case["t0"], case["fs"], cinv_f32(case["c"]),
                     VS=case["VS"], DV=case["DV"], interp=interp, fmod=fmod).reshape(-1, order="F")     # feval: I x 1 x 1, I1 fastest
    tol = 2e-3 if prec == "halfT" else (1e-2 if interp == "nearest" else 2e-5 if not fmod else 2e-4)
    assert rel_err(ys[1].reshape(-1), ref) <= tol
    assert rel_err(ys[1], ys[0]) <= 1e-6                                     # same arithmetic as the prebuilt instantiation


@pytest.mark.gpu
def test_jit_modes_syn_mul_and_pixel_weights(tmp_path, monkeypatch):
    """the specialised kernel also serves kept dimensions (planes) and a pixel x receiver apodization"""
    import torch
    from qups_amd import das_spec
    monkeypatch.setenv("QDAS_CACHE_DIR", str(tmp_path))
    case = make_case(seq="PW", interp="cubic", seed=33, N=12, M=10, I1=100, I2=20)
    args = (case["Pi"], case["Pr"], case["Pv"], case["Nv"], torch.from_numpy(case["x"]), case["t0"], case["fs"], case["c"])
    rng = np.random.default_rng(5)
    a_pix = rng.random((100, 20, 1, 12, 1)).astype(np.float32)
    for fun, ap in (("SYN", None), ("MUL", None), ("DAS", a_pix)):
        extra = ["apod", ap] if ap is not None else []
        y0, p0 = das_spec(fun, *args, *case["opt"], "interp", "cubic", *extra, return_plan=True, kernel=2)
        y1, p1 = das_spec(fun, *args, *case["opt"], "interp", "cubic", *extra, return_plan=True, kernel=2, jit=True)
        assert "[jit " in p1.kernel_name() and "[prebuilt]" in p0.kernel_name()
        assert rel_err(y1.cpu().numpy(), y0.cpu().numpy()) <= 1e-6, fun
        p0.close(); p1.close()


@pytest.mark.gpu
def test_environment_default_builds_the_specialised_kernel(tmp_path, monkeypatch):
    """QDAS_JIT=1: every das_spec / UltrasoundSystem call asks for the plan-specialised kernel (as the reference compiles its kernel per call)"""
    import torch
    from qups_amd import das_spec
    from tests.cases import make_case
    monkeypatch.setenv("QDAS_CACHE_DIR", str(tmp_path))
    case = make_case(seq="PW", interp="linear", seed=3, N=16, M=8, I1=64, I2=8)
    args = (case["Pi"], case["Pr"], case["Pv"], case["Nv"], torch.from_numpy(case["x"]), case["t0"], case["fs"], case["c"])
    y0, p0 = das_spec("DAS", *args, *case["opt"], "interp", "linear", return_plan=True)
    monkeypatch.setenv("QDAS_JIT", "1")
    y1, p1 = das_spec("DAS", *args, *case["opt"], "interp", "linear", return_plan=True)
    assert "[prebuilt]" in p0.kernel_name() and "[jit " in p1.kernel_name()
    a, b = y0.cpu().numpy(), y1.cpu().numpy()
    assert np.abs(a - b).max() <= 2e-6 * np.abs(a).max()
