"""world_size-2/3 gloo tests of the multi-GPU layout (pixel slabs + one all_gather) on CPU: the local
compute is the oracle restricted to the rank's slab, so what is exercised is exactly the N > 1 host logic
of qups_amd.dist (shard ranges, ragged slabs, plane layouts of the keep_rx/keep_tx modes, the collective)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.cases import cinv_f32, make_case


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, cases, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import das_oracle as O
        from qups_amd import build_problem, parse_options
        from qups_amd.dist import ShardedDasPlan
        for fun, I1, I2, *rest in cases:
            mirror_slabs = bool(rest and rest[0])
            balance = rest[1] if len(rest) > 1 else None
            case = make_case(seq="PW", interp="linear", seed=3, N=4, M=3, I1=I1, I2=I2)
            x = torch.from_numpy(case["x"])
            opts = parse_options(x, list(case["opt"]) + ["interp", "linear"])
            prob = build_problem(fun, case["Pi"], case["Pr"], case["Pv"], case["Nv"], x.shape, case["t0"], case["fs"], case["c"], opts)
            full = O.das_spec(fun, case["Pi"], case["Pr"], case["Pv"], case["Nv"], case["x"], case["t0"], case["fs"],
                              cinv_f32(case["c"]), VS=case["VS"], DV=case["DV"], interp="linear")       # I1 x I2 x 1 x oN x oM
            oN, oM = prob.osize
            full_cm = torch.from_numpy(np.ascontiguousarray(full.reshape(prob.I, oN, oM, order="F").transpose(2, 1, 0)).astype(np.complex64))[None]

            def compute(xc, F, b, c, full_cm=full_cm):   # the slab a GPU rank would produce: (1, oM, oN, count)
                return full_cm[..., b:b + c].contiguous()

            plan = ShardedDasPlan(prob, rank, world, compute=compute, mirror_slabs=mirror_slabs, balance=balance, balance_granule=1)
            y = plan.execute_colmajor(x.permute(2, 1, 0).contiguous(), 1)
            ok = bool(torch.equal(y, full_cm)) and tuple(y.shape) == (1, oM, oN, prob.I) and plan.mirror_slabs == mirror_slabs
            q.put((rank, fun, I1 * I2, ok, plan.i_begin, plan.i_count))
    finally:
        dist.destroy_process_group()


CASES = [("DAS", 16, 4), ("DAS", 9, 3), ("SYN", 7, 3), ("BF", 6, 2), ("DAS", 1, 1)]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_gather_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, CASES, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world * len(CASES))]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[3] for r in res), res
    for fun, I1, I2 in CASES:
        spans = sorted((b, c) for _, f, I, _, b, c in res if f == fun and I == I1 * I2)
        assert spans[0][0] == 0 and sum(c for _, c in spans) == I1 * I2
        for (b0, c0), (b1, _) in zip(spans, spans[1:]):
            assert b0 + c0 == b1


def test_shard_range_properties():
    from qups_amd.dist import shard_range
    for I in (0, 1, 7, 64, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [shard_range(I, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == I
            assert all(b0 + c0 == b1 for (b0, c0), (b1, _) in zip(spans, spans[1:]))
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _tx_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import das_oracle as O
        from qups_amd.dist import das_spec_tx_sharded, shard_range
        for seq, fun, tvec in (("PW", "DAS", False), ("FSA", "DAS", True), ("DV", "SYN", False)):
            case = make_case(seq=seq, interp="cubic", seed=5, N=6, M=7, I1=20, I2=5)
            M = case["M"]
            rng = np.random.default_rng(9)
            t0 = case["t0"]
            if tvec:
                t0 = (case["t0"] + 1.0 / case["fs"] * rng.integers(-2, 3, (1, 1, M))).astype(np.float64)
            am = rng.uniform(0.2, 1, (1, 1, 1, 1, M))
            an = rng.uniform(0.2, 1, (1, 1, 1, case["N"], 1))
            ora = lambda f, Pi, Pr, Pv, Nv, x, t0_, fs, c, *opts: torch.from_numpy(np.ascontiguousarray(O.das_spec(
                f, Pi, Pr, Pv, Nv, x, t0_, fs, cinv_f32(c), VS=case["VS"], DV=case["DV"], interp="cubic",
                apod=tuple(opts[k + 1] for k in range(len(opts)) if isinstance(opts[k], str) and opts[k] == "apod"))))
            full = ora(fun, case["Pi"], case["Pr"], case["Pv"], case["Nv"], case["x"], t0, case["fs"], case["c"], "apod", am, "apod", an)
            b, c = shard_range(M, rank, world)
            y = das_spec_tx_sharded(fun, case["Pi"], case["Pr"], case["Pv"], case["Nv"], case["x"][:, :, b:b + c], t0, case["fs"], case["c"],
                                    *case["opt"], "interp", "cubic", "apod", am, "apod", an, rank=rank, world=world, M=M, compute=ora)
            err = float((y - full).abs().max() / full.abs().max())
            q.put((rank, seq, fun, err, tuple(y.shape) == tuple(full.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_transmit_sharded_allreduce_gloo(world):
    """the alternative multi-GPU layout: every rank beamforms all pixels over ITS transmits, one all_reduce sums the partial images"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tx_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world * 3)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[4] and r[3] <= 1e-12 for r in res), res


MIRROR_CASES = [("DAS", 16, 4, True), ("DAS", 5, 10, True), ("DAS", 3, 2, True), ("DAS", 7, 12, True)]


@pytest.mark.parametrize("world", [2, 3])
def test_mirror_slab_gather_gloo(world):
    """the mirror-slab layout (rank r: columns [c0, c1) of the first half AND their mirror images, gathered as [A | B] pairs and laid out as
    A_0 .. A_{G-1} B_{G-1} .. B_0): shard ranges, ragged and empty ranks (fewer half-columns than ranks), the single collective"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, MIRROR_CASES, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world * len(MIRROR_CASES))]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[3] for r in res), res
    for fun, I1, I2, _ in MIRROR_CASES:
        spans = sorted((b, c) for _, f, I, _, b, c in res if f == fun and I == I1 * I2)
        assert spans[0][0] == 0 and sum(c for _, c in spans) == I1 * I2 // 2            # the slabs A tile the first half of the columns
        for (b0, c0), (b1, _) in zip(spans, spans[1:]):
            assert b0 + c0 == b1 and b0 % I1 == 0


BALANCED_CASES = [("DAS", 9, 10, True, [5.0, 1.0, 1.0, 1.0, 1.0]),                      # 5 half-columns, the outermost five times as expensive: ragged at 2 and 3 ranks
                  ("DAS", 4, 16, True, [3.0, 1.0]),                                     # a per-BLOCK profile (2 blocks over 8 half-columns)
                  ("DAS", 6, 4, True, [1.0, 1.0]),                                      # fewer half-columns than ranks at world 3
                  ("DAS", 5, 14, True, "measure")]                                      # rank 0 "measures" (no device here: no profile) -> equal widths, agreed by broadcast


@pytest.mark.parametrize("world", [2, 3])
def test_balanced_mirror_slabs_gloo(world):
    """mirror slabs of equal COST instead of equal width (``ShardedDasPlan(balance=...)``, ``balanced_column_bounds``): the ranks derive the same
    ragged column boundaries from the same profile, the padded all_gather lays the unequal slabs out as A_0 .. A_{G-1} B_{G-1} .. B_0 -- the image is the whole-image
    plan's bit for bit"""
    from qups_amd.dist import balanced_column_bounds, expand_block_cost
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, BALANCED_CASES, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world * len(BALANCED_CASES))]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[3] for r in res), res
    for fun, I1, I2, _, cost in BALANCED_CASES:
        spans = sorted((b, c) for _, f, I, _, b, c in res if f == fun and I == I1 * I2)
        assert spans[0][0] == 0 and sum(c for _, c in spans) == I1 * I2 // 2
        for (b0, c0), (b1, _) in zip(spans, spans[1:]):
            assert b0 + c0 == b1 and b0 % I1 == 0
        if not isinstance(cost, str):
            want = balanced_column_bounds(expand_block_cost(cost, I2 // 2), world)
            assert [b // I1 for b, _ in spans] == want[:-1], (spans, want)
    # the expensive outer column got a rank of its own; the profile changed the layout
    first = sorted((b, c) for _, f, I, _, b, c in res if I == 90)
    assert first[0][1] == 9 and len({c for _, c in first}) > 1, first


def test_balanced_column_bounds_properties():
    from qups_amd.dist import balanced_column_bounds, expand_block_cost
    rng = np.random.default_rng(0)
    for _ in range(500):
        h, w = int(rng.integers(0, 40)), int(rng.integers(1, 10))
        c = rng.uniform(0, 1, h) ** 3
        b = balanced_column_bounds(c, w)
        assert len(b) == w + 1 and b[0] == 0 and b[-1] == h and all(b[i] <= b[i + 1] for i in range(w))
        if h >= w:
            assert all(b[i] < b[i + 1] for i in range(w))               # nobody idle while there are columns
    # a smooth profile is balanced to within one column's cost
    c = 1 + 0.15 * np.linspace(1, 0, 512)
    b = balanced_column_bounds(c, 8)
    share = [c[b[i]:b[i + 1]].sum() for i in range(8)]
    assert max(share) - min(share) <= 2 * c.max() and b != [64 * k for k in range(9)]
    assert balanced_column_bounds(np.ones(512), 8) == [64 * k for k in range(9)]       # a flat profile: equal widths
    assert balanced_column_bounds([np.nan, 1.0], 2) == [0, 1, 2] and balanced_column_bounds([], 3) == [0, 0, 0, 0]
    assert np.allclose(expand_block_cost([2, 1], 5), [1, 1, 1 / 3, 1 / 3, 1 / 3])
    e = expand_block_cost([1, 2, 3, 4, 5], 2)              # more blocks than columns: nothing of the profile is dropped
    assert np.isclose(e.sum(), 15.0) and e.shape == (2,) and np.all(e > 0)
    # boundaries snapped to the kernel's tile width: no rank gets a partial column tile (a 266-column slab of C3 costs two tile rounds, a 256-column one a single round)
    bq = balanced_column_bounds(c, 8, 32)
    assert all(v % 32 == 0 for v in bq) and bq[0] == 0 and bq[-1] == 512 and all(bq[i] < bq[i + 1] for i in range(8))
    assert balanced_column_bounds(np.ones(100), 3, 32) == [0, 32, 64, 100]


def _fold_np(x):
    """numpy restatement of csrc/fold.hip on a column-major frame [m, n, t]: xs[m, n] = x[m, n] + x[n, m] for n < m, xs[m, m] = x[m, m], nothing below"""
    M, N, T = x.shape
    xs = np.zeros_like(x)
    for m in range(M):
        for n in range(m + 1):
            xs[m, n] = x[m, n] + (x[n, m] if n != m else 0)
    return xs


def _folded_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import das_oracle as O
        from qups_amd import build_problem, parse_options
        from qups_amd.dist import FoldedReplicator, ShardedDasPlan
        case = make_case(seq="FSA", interp="linear", seed=5, N=6, I1=10, I2=4)
        xfull = case["x"]                                           # T x N x M on every rank; only the acquisition rank may look at it
        T, N, M = xfull.shape
        full = O.das_spec("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], xfull, case["t0"], case["fs"], cinv_f32(case["c"]),
                          VS=case["VS"], DV=case["DV"], interp="linear").reshape(-1, order="F")
        x = torch.from_numpy(xfull)
        prob = build_problem("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], x.shape, case["t0"], case["fs"], case["c"],
                             parse_options(x, list(case["opt"]) + ["interp", "linear"]))

        def compute(xs_cm, F, b, c):                                # the slab from the FOLDED frame: the oracle over it sums the upper triangle = every pair
            xs = xs_cm.numpy().transpose(2, 1, 0)                   # T x N x M, zeros where n > m
            y = O.das_spec("DAS", case["Pi"], case["Pr"], case["Pv"], case["Nv"], xs, case["t0"], case["fs"], cinv_f32(case["c"]),
                           VS=case["VS"], DV=case["DV"], interp="linear").reshape(-1, order="F")
            return torch.from_numpy(y[b:b + c].astype(np.complex64)).reshape(1, 1, 1, c)

        rep = FoldedReplicator(N, T, "cpu", src=0, fold=lambda xc: torch.from_numpy(_fold_np(xc.numpy())))
        plan = ShardedDasPlan(prob, rank, world, compute=compute, mirror_slabs=False)
        oks = []
        for f in range(3):                                          # a stream: the transfer of frame f + 1 is started before frame f is beamformed
            frames = [(xfull * (1 + f)).astype(np.complex64), None]
            xc = torch.from_numpy(np.ascontiguousarray(frames[0].transpose(2, 1, 0))) if rank == 0 else None
            slot, work = rep.send(xc, rank, async_op=True)
            xs = rep.receive(slot, rank, work)
            y = plan.execute_colmajor(xs, 1).reshape(-1).numpy()
            oks.append(bool(np.abs(y - (1 + f) * full).max() <= 2e-6 * np.abs(full).max() * (1 + f)))
            # nothing below the diagonal arrived, and exactly the triangle's bytes travelled
            low = xs.numpy()[np.triu_indices(N, 1)]
            oks.append(not low.any() and rep.bytes_per_frame == N * (N + 1) // 2 * T * 8)
        q.put((rank, all(oks)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_folded_replication_gloo(world):
    """a reciprocal frame travels FOLDED from the acquisition rank (half the bytes: the packed upper triangle, one broadcast), every rank beamforms its
    slab from the folded frame: the image is the full N x M sum (linearity of the interpolators)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_folded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res
