"""Build check (CPU only): register metadata of the kernels `make` produced, read from the code objects embedded in
qups_amd/libqdas.so (tools/kernel_regs.py).  The tiled kernel is sized for 4 waves per SIMD (128 VGPRs); a spilled register there
means scratch traffic in the stage loop (round 1: 170-230 spilled VGPRs in every general instantiation) -- never again silently."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def table():
    import kernel_regs
    so = os.path.join(ROOT, "qups_amd", "libqdas.so")
    if not os.path.exists(so) or not os.path.exists(kernel_regs.READELF):
        pytest.skip("libqdas.so / llvm-readelf not available")
    rows = kernel_regs.kernel_table(so)
    assert rows, "no gfx950 code objects found in libqdas.so"
    return rows


def _is_probe(demangled: str) -> bool:
    """das_tile_kernel<INTERP, ST, FMOD, WTAB, SYM, FB2, FB4, WAVES, MB, W, NBUF, PSZ, BPC, PROBE, ...>: template argument 14"""
    import re
    args = re.sub(r"HIP_vector_type<(\w+), 2u>", r"\1x2", demangled[demangled.index("<") + 1:]).split(",")
    return args[13].strip() == "true"


def test_tiled_kernels_do_not_spill(table):
    tiled = [r for r in table if "das_tile_kernel" in r["name"]]
    # libqdas.so carries exactly the variants das_tile_cfg.h TILE_PREBUILT names (the rest is built on demand: tests/test_jit.py) plus the probes
    from qups_amd import _lib, warm
    L = _lib.lib()
    want = sum(1 for v in warm.all_variants() if L.qdas_kernel_variant_prebuilt(*v) == 1)
    import kernel_regs
    names = kernel_regs.demangle([r["name"] for r in tiled])
    full = [n for n in names if not _is_probe(n)]
    assert 30 <= want <= 80 and len(full) == want, (want, len(full))
    bad = [(r["name"][:90], r["vgpr_spill"], r["scratch"]) for r in tiled if r["vgpr_spill"] or r["scratch"]]
    assert not bad, bad
    assert max(r["vgpr"] + r["agpr"] for r in tiled) <= 128      # 16 waves per CU = 4 per SIMD


def test_no_kernel_spills_vector_registers_or_uses_scratch(table):
    bad = [(r["name"][:90], r["vgpr_spill"], r["scratch"]) for r in table if r["vgpr_spill"] or r["scratch"]]
    assert not bad, bad
