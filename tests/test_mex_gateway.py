"""mex/qdas_mex.c, the MATLAB binding of the C ABI, checked without MATLAB: compiled against tests/fake_mex/ (a stand-in that
declares only the mx / mex prototypes the gateway uses -- a syntax / ABI check, NOT a MATLAB), and, on a GPU box, driven through
create / execute / info / destroy, multi-device plans, the stateless commands of the other launch sites (delays, lut, greens, convd, hilbert) and its error paths over a malloc-backed fake runtime and the real libqdas.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_mex")
SRC = os.path.join(ROOT, "mex", "qdas_mex.c")
CFLAGS = ["-std=c99", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-I", os.path.join(ROOT, "include"), "-I", FAKE]


@pytest.mark.parametrize("gpu", [False, True])
def test_gateway_compiles_against_the_declared_mx_api(gpu, tmp_path):
    cmd = ["gcc", "-c", *CFLAGS, *( ["-DQDAS_MEX_GPU"] if gpu else []), SRC, "-o", str(tmp_path / "qdas_mex.o")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.gpu
def test_gateway_runs_over_the_fake_runtime(tmp_path):
    exe = str(tmp_path / "run_gateway")
    lib = os.path.join(ROOT, "qups_amd")
    cmd = ["gcc", "-O1", *CFLAGS, SRC, os.path.join(FAKE, "fake_mex_runtime.c"), os.path.join(FAKE, "run_gateway.c"),
           "-L", lib, "-lqdas", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # (the driver compares multi-device plans -- pixel slabs, plain kernel -- with the one-device plan bit for bit: the lateral-mirror mode of
    #  a whole-image plan sums the mirrored half in another order, so it is switched off for this comparison)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, QDAS_NO_MIRROR="1"))
    assert r.returncode == 0 and "fake-MEX gateway OK" in r.stdout, r.stdout + r.stderr
    # (the launch sites beside DAS -- delays, wsinterpd2 as bfDASLUT reaches it, greens, convd, hilbert -- each driven through the gateway
    #  and compared bit for bit with the C ABI called directly on device arrays)
    assert "delays / lut / greens / convd / hilbert / wsinterpd / shiftsum through the gateway: bit-identical to the C ABI" in r.stdout, r.stdout
