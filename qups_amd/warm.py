"""Build variants of the fused kernel ahead of time: ``python -m qups_amd.warm [--jobs J] [--all | census files ...]``.

``libqdas.so`` carries the instantiations the BASELINE configurations and frame streams launch (``csrc/das_tile_cfg.h`` ``TILE_PREBUILT``);
every other point of the template's matrix is compiled by hiprtc the first time a plan needs it (about 2 s, then cached under
``$QDAS_CACHE_DIR`` or ``~/.cache/qdas``) -- the reference compiles ALL its kernels that way, per ``UltrasoundSystem``
(``src/UltrasoundSystem.m:5527-5625`` ``recompileCUDA``).  This module fills that cache for a list of variants with one compiler process per
core, so that a deployment or a test session never waits inside ``qdas_plan_create``: the list is every variant (``--all``), or the census of a
workload (files written under ``QDAS_KERNEL_CENSUS=<file>``: ``tools/kernel_census.py``).  No device is needed.
"""
from __future__ import annotations

import os
import subprocess
import sys
from typing import Iterable, List, Tuple

Variant = Tuple[int, int, int, int]          # (launch configuration, interpolator flag, remodulation, weight table)

CONFIGS = (0, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14, 16, 17, 18, 19, 20, 21)
INTERPS = (0, 1, 2, 3, 5)


def all_variants() -> List[Variant]:
    return [(c, i, f, w) for c in CONFIGS for i in INTERPS for f in (0, 1) for w in (0, 1)]


def read_census(paths: Iterable[str]) -> List[Variant]:
    """variants named by kernel-census files (lines ``ci interp sample_bytes fm wt probe``; probes are always prebuilt)"""
    out = set()
    for path in paths:
        with open(path) as fh:
            for line in fh:
                p = line.split()
                if len(p) == 6 and p[5] == "0":
                    out.add((int(p[0]), int(p[1]), int(p[3]), int(p[4])))
    return sorted(out)


def build(variants: Iterable[Variant]):
    """build in THIS process; returns (built, prebuilt, unknown, failures)"""
    from . import _lib
    L = _lib.lib()
    built = pre = unknown = 0
    failures = []
    for c, i, f, w in variants:
        rc = L.qdas_kernel_variant_build(c, i, f, w)
        if rc == 0:
            built += 1
        elif rc == 2:
            pre += 1
        elif rc == 3:
            unknown += 1
        else:
            failures.append(((c, i, f, w), (L.qdas_last_error() or b"").decode(errors="replace")))
    return built, pre, unknown, failures


def warm(variants: Iterable[Variant], jobs: int | None = None, quiet: bool = True) -> int:
    """build ``variants`` with ``jobs`` compiler processes (default: the cores of this host, at most 32); returns the number of failures"""
    vs = sorted(set(variants))
    if not vs:
        return 0
    jobs = max(1, min(jobs or min(os.cpu_count() or 4, 32), len(vs)))
    procs = []
    for j in range(jobs):
        part = vs[j::jobs]
        arg = ";".join(",".join(map(str, v)) for v in part)
        procs.append(subprocess.Popen([sys.executable, "-m", "qups_amd.warm", "--worker", arg],
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                                      cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    bad = 0
    for p in procs:
        out, _ = p.communicate()
        if p.returncode:
            bad += 1
        if not quiet or p.returncode:
            sys.stderr.write(out)
    return bad


_SPEC_WORKER = r"""
import ctypes as C, os, sys
sys.path.insert(0, sys.argv[1])
from qups_amd import _lib
L = _lib.lib()
f = L.qdas_debug_jit_compile
f.argtypes = [C.c_int] * 4 + [C.c_ulonglong] * 3 + [C.c_char_p, C.c_size_t, C.POINTER(C.c_ulonglong)]
bad = 0
for spec in sys.argv[2:]:
    os.environ["QDAS_JIT_DEBUG_SPEC"] = spec
    msg, n = C.create_string_buffer(4000), C.c_ulonglong()
    if f(0, 0, 0, 0, 0, 0, 0, msg, 4000, C.byref(n)):
        print("FAILED", spec[:120], msg.value.decode(errors="replace")[:400])
        bad += 1
sys.exit(1 if bad else 0)
"""


def read_specs(paths: Iterable[str]) -> List[str]:
    """complete JitSpecs (lines ``raw:name=value,...``: what ``QDAS_JIT_SPEC_LOG=<file>`` appends per plan-specialised build)"""
    out = set()
    for path in paths:
        with open(path) as fh:
            out.update(ln.strip() for ln in fh if ln.startswith("raw:"))
    return sorted(out)


def warm_specs(specs: Iterable[str], cache_dir: str, jobs: int | None = None, quiet: bool = True) -> int:
    """Build plan-specialised (hiprtc) kernels from their logged specs into ``cache_dir``, one compiler process per core: a later process finds them there through
    ``QDAS_CACHE_DIR`` or -- read-only, copied into its own cache on a hit -- ``QDAS_JIT_WARM_DIR`` (csrc/jit.hip).  No device needed.  Returns the number of failed workers."""
    sp = sorted(set(specs))
    if not sp:
        return 0
    os.makedirs(cache_dir, mode=0o700, exist_ok=True)
    jobs = max(1, min(jobs or min(os.cpu_count() or 4, 64), len(sp)))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, QDAS_CACHE_DIR=cache_dir)
    for k in ("QDAS_JIT_DEFINES", "QDAS_JIT_FLAGS", "QDAS_JIT_MB", "QDAS_JIT_W", "QDAS_JIT_NBUF", "QDAS_JIT_WARM_DIR", "QDAS_JIT_SPEC_LOG"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, "-c", _SPEC_WORKER, root] + sp[j::jobs], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for j in range(jobs)]
    bad = 0
    for p in procs:
        out, _ = p.communicate()
        if p.returncode:
            bad += 1
        if not quiet or p.returncode:
            sys.stderr.write(out)
    return bad


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    if argv[:1] == ["--worker"]:
        vs = [tuple(int(t) for t in item.split(",")) for item in argv[1].split(";") if item]
        built, pre, unknown, failures = build(vs)
        print(f"qups_amd.warm: {built} built or cached, {pre} prebuilt, {unknown} not in the matrix, {len(failures)} failed")
        for v, why in failures:
            print(f"  {v}: {why[:300]}")
        return 1 if failures else 0
    jobs = None
    if "--jobs" in argv:
        k = argv.index("--jobs")
        jobs = int(argv[k + 1])
        del argv[k:k + 2]
    vs = all_variants() if (not argv or argv == ["--all"]) else read_census(argv)
    return warm(vs, jobs, quiet=False)


if __name__ == "__main__":
    sys.exit(main())
