import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()


def pytest_collection_finish(session):
    """GPU sessions: the variants of the fused kernel that this suite launches and libqdas.so does not carry (tests/suite_kernels.txt: the census of
    a full run, tools/kernel_census.py) are built ahead of the tests, one compiler process per core (qups_amd/warm.py) -- instead of one by one,
    ~2 s each, inside the first plan that needs them.  A stale list only means that a variant is built on demand after all."""
    if os.environ.get("QDAS_NO_WARM") or not any(item.get_closest_marker("gpu") for item in session.items):
        return
    xw = os.environ.get("PYTEST_XDIST_WORKER")               # pytest -n K (the soak runs): every worker collects; one of them warms the shared variant cache,
    if xw and xw != "gw0":                                   # the others start at once (K copies of the warm-up were ~9 minutes of a 12-worker soak in round 6)
        return
    try:
        import torch
        if not torch.cuda.is_available():
            return
        import time
        if not os.environ.get("QDAS_CACHE_DIR") and not os.environ.get("HOME"):      # (no private place for the disk cache: make one for this session)
            import tempfile
            os.environ["QDAS_CACHE_DIR"] = tempfile.mkdtemp(prefix="qdas_cache_")
        # the variants live in ONE directory for the whole session, whatever scratch directory a test points QDAS_CACHE_DIR at (csrc/jit.hip cache_dir)
        os.environ.setdefault("QDAS_VARIANT_CACHE_DIR", os.environ.get("QDAS_CACHE_DIR") or os.path.join(os.environ["HOME"], ".cache", "qdas"))
        from qups_amd import warm
        t = time.perf_counter()
        vs = warm.read_census([os.path.join(ROOT, "tests", "suite_kernels.txt")])
        bad = warm.warm(vs)
        sys.stderr.write(f"[conftest] kernel cache warmed: {len(vs)} variants listed, {bad} worker(s) failed, {time.perf_counter() - t:.1f} s\n")
        # the PLAN-SPECIALISED builds of the suite (tests/suite_jit_kernels.txt: the QDAS_JIT_SPEC_LOG of a full run; ~3 s of hiprtc each, one by one inside the tests:
        # a quarter of round 5's suite time) go into a read-only warm directory the same way (csrc/jit.hip QDAS_JIT_WARM_DIR: a hit is copied into the cache directory
        # the test watches, as if it had been built there)
        jl = os.path.join(ROOT, "tests", "suite_jit_kernels.txt")
        if os.path.exists(jl) and not os.environ.get("QDAS_NO_JIT_WARM") and not xw:      # (the warm directory is per process: not under xdist)
            import tempfile
            t = time.perf_counter()
            wd = tempfile.mkdtemp(prefix="qdas_jit_warm_")
            specs = warm.read_specs([jl])
            bad = warm.warm_specs(specs, wd)
            os.environ["QDAS_JIT_WARM_DIR"] = wd
            sys.stderr.write(f"[conftest] hiprtc builds warmed: {len(specs)} specs, {bad} worker(s) failed, {time.perf_counter() - t:.1f} s\n")
    except Exception as ex:                                  # (never fatal: the variants are then built on demand)
        sys.stderr.write(f"[conftest] kernel cache not warmed: {ex}\n")
