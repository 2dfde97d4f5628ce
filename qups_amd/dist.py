"""Multi-GPU execution of the DAS path: one process per GPU, pixels sharded, one gather.

Every output pixel is independent (reference ``src/bf.cu:85-141`` has no cross-pixel state), so the image
is split into ``world`` contiguous slabs of the linear pixel index.  Each rank
beamforms its slab from its own replica of the channel data and geometry; the slabs are then
concatenated with ONE collective -- ``all_gather`` over RCCL/xGMI (backend ``"nccl"`` on ROCm), or gloo
in the CPU tests.  There is no other communication on this path; the reference has no multi-GPU path at
all (one ``gpuDevice`` per MATLAB process, ``README.md:232``).
"""
from __future__ import annotations

from typing import Callable, Tuple


def shard_range(I: int, rank: int, world: int) -> Tuple[int, int]:
    """``(i_begin, i_count)`` of ``rank``'s slab; slabs are contiguous, disjoint and cover ``[0, I)``."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    b, e = I * rank // world, I * (rank + 1) // world
    return b, e - b


def gather_pixels(y_local, I: int, world: int, group=None):
    """All-gather slabs ``(..., count_r)`` (pixel index LAST, i.e. the column-major parent of an
    ``I x [N] x [M] x F`` array) into ``(..., I)`` on every rank.

    Equal slabs use one ``all_gather_into_tensor``; ragged slabs are padded to the largest slab so that
    it is still a single collective.
    """
    import torch
    import torch.distributed as dist

    if world == 1:
        return y_local
    lead = tuple(y_local.shape[:-1])
    ax = len(lead)                                                        # the pixel axis
    counts = [shard_range(I, r, world)[1] for r in range(world)]
    cmax = max(counts)
    cplx = y_local.is_complex()
    src = torch.view_as_real(y_local) if cplx else y_local               # collectives move (re, im) pairs of the real type
    if all(c == cmax for c in counts) and not lead_has_planes(lead):
        out = torch.empty((I,) + tuple(src.shape[ax + 1:]), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(out, src.reshape((cmax,) + tuple(src.shape[ax + 1:])).contiguous(), group=group)
        full = out.reshape(lead + (I,) + tuple(src.shape[ax + 1:]))
    else:
        pad = torch.zeros(lead + (cmax,) + tuple(src.shape[ax + 1:]), dtype=src.dtype, device=src.device)
        pad.narrow(ax, 0, src.shape[ax]).copy_(src)
        flat = pad.reshape((1,) + tuple(pad.shape)).contiguous()            # concatenated along a new dim 0
        out = torch.empty((world,) + tuple(pad.shape), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(out, flat, group=group)
        full = torch.cat([out[r].narrow(ax, 0, counts[r]) for r in range(world)], dim=ax)
    return torch.view_as_complex(full.contiguous()) if cplx else full


def lead_has_planes(lead) -> bool:
    """True when the slab carries more than one (frame, n, m) plane: then slabs do not concatenate flat."""
    n = 1
    for s in lead:
        n *= int(s)
    return n > 1


def mirror_slab_columns(I2: int, rank: int, world: int) -> Tuple[int, int]:
    """columns ``[c0, c1)`` of the FIRST half of an even number of columns that rank ``rank`` takes as its slab A; its slab B is the
    mirror image ``[I2 - c1, I2 - c0)``"""
    h = I2 // 2
    return h * rank // world, h * (rank + 1) // world


def balanced_column_bounds(cost, world: int, granule: int = 1):
    """Column boundaries ``b[0] = 0 <= b[1] <= ... <= b[world] = len(cost)`` that give every rank (nearly) the same share of ``sum(cost)``:
    ``cost[c]`` = what column ``c`` of the first half of the image (and its mirror image) costs to beamform -- measured per column block by
    :func:`measure_column_cost`, or any model.  Equal column counts (:func:`mirror_slab_columns`) leave the rank with the outermost columns the
    slowest: the delay gradient -- and with it the spread of the LDS gathers -- grows away from the array (C3 at 8 ranks: 2.65 ms against
    2.36 ms, ``profiles/r04/slab_kernel_times_c3.txt``).  Greedy on the cumulative cost at column granularity; every rank gets at least one column
    while there are columns left; identical on every rank that passes the same ``cost`` (pure, deterministic).

    ``granule``: boundaries are multiples of it.  The fused kernel works in TILES of 16-128 columns and a rank's kernel time is set by the number of
    tile ROUNDS its slab needs on the device's CUs, not by its column count: at C3 a slab of 266 columns (9 column tiles x 32 depth tiles = 288 workgroups on
    256 CUs: two rounds) takes 13.4 ms where 256 columns take 7.2 ms (``profiles/r05/slab_kernel_times_c3.txt``) -- pass the plans' tile width
    (``DasPlan.tile_shape()[1]``) so that no rank gets a partial column tile."""
    import numpy as np
    c = np.asarray(cost, dtype=np.float64).reshape(-1)
    if granule > 1 and c.size:
        ng = (c.size + granule - 1) // granule
        cg = np.array([c[g * granule:(g + 1) * granule].sum() for g in range(ng)])
        bg = balanced_column_bounds(cg, world, 1)
        return [min(int(b) * granule, int(c.size)) for b in bg]
    h = int(c.size)
    if world < 1:
        raise ValueError("world must be >= 1")
    if h == 0:
        return [0] * (world + 1)
    if not np.all(np.isfinite(c)) or np.any(c < 0) or c.sum() <= 0:
        c = np.ones(h)
    cum = np.concatenate([[0.0], np.cumsum(c)])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(cum, target, side="left"))             # first boundary whose cumulative cost reaches the target ...
        if k > 0 and abs(cum[k - 1] - target) <= abs(cum[min(k, h)] - target):
            k -= 1                                                       # ... or the one before it, whichever is closer
        k = max(k, min(bounds[-1] + 1, h))                              # rank r-1 gets at least one column while there are columns
        if h >= world:
            k = min(k, h - (world - r))                                  # ... and so does every rank that follows
        bounds.append(min(k, h))
    bounds.append(h)
    return bounds


def expand_block_cost(block_cost, ncols: int):
    """per-column cost from a per-BLOCK measurement (``len(block_cost)`` equal blocks over ``ncols`` columns): every column of a block takes its block's
    cost per column"""
    import numpy as np
    b = np.asarray(block_cost, dtype=np.float64).reshape(-1)
    nb = b.size
    out = np.zeros(ncols)
    if ncols <= 0 or nb == 0:
        return out
    if nb > ncols:                                  # more blocks than columns: a column takes the SUM of the blocks that fall on it (nothing of the profile is dropped)
        for k in range(nb):
            out[min(ncols - 1, k * ncols // nb)] += b[k]
        return out
    edges = [ncols * k // nb for k in range(nb + 1)]
    for k in range(nb):
        w = edges[k + 1] - edges[k]
        out[edges[k]:edges[k + 1]] = b[k] / w       # (nb <= ncols: every block has at least one column)
    return out


def measure_column_cost(prob, device=None, nblocks: int = 8, reps: int = 3, fixed_ms: float | None = None, **plan_kw):
    """Kernel time [ms] of the mirror-slab plan of each of ``nblocks`` equal column blocks of the first half of the image, on THIS device, minus the
    per-launch cost that does not depend on the slab (``fixed_ms``; default: estimated as the part of the times that does not scale with the block --
    the reciprocity fold pass of the whole frame above all).  The channel data are zeros: the kernels' time does not depend on the data.  Costs a few
    plan creations (~0.1 s each once the kernel is built); meant to run once per geometry, on one rank, and be broadcast (``ShardedDasPlan(balance=...)``)."""
    import numpy as np
    import torch
    from .das_spec import DasPlan, _data_dtype
    I1, I2, _ = prob.Isz
    h = I2 // 2
    nblocks = max(1, min(nblocks, h))
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    xc = torch.zeros((1, prob.M, prob.N, prob.T), dtype=_data_dtype(prob.prec), device=dev)
    times = []
    for k in range(nblocks):
        c0, c1 = h * k // nblocks, h * (k + 1) // nblocks
        with DasPlan(prob, device=dev, i_begin=c0 * I1, i_count=(c1 - c0) * I1, mirror_slab=True, **plan_kw) as plan:
            plan.set_timing(True)
            ks = []
            for _ in range(reps + 1):
                plan.execute_colmajor(xc, 1)
                ks.append(plan.last_kernel_ms())
            times.append(float(np.median(ks[1:])))
    t = np.asarray(times)
    if fixed_ms is None:
        # blocks have (nearly) equal column counts: what they share is bounded by the cheapest block; half of it is a conservative estimate of the fixed part
        fixed_ms = 0.5 * float(t.min())
    return np.maximum(t - fixed_ms, 1e-3 * float(t.max())), t


def gather_mirror_slabs(y_local, I1: int, I2: int, world: int, group=None, bounds=None):
    """All-gather the ranks' ``[slab A | slab B]`` outputs (``(..., 2 * count_r)``, one plane) into the image ``(..., I1 * I2)``: one collective
    (padded to the largest rank), then the slabs A in rank order followed by the slabs B in REVERSE rank order.  ``bounds``: the ranks' column
    boundaries in the first half (``balanced_column_bounds``); default: equal column counts."""
    import torch
    import torch.distributed as dist

    cols = [mirror_slab_columns(I2, r, world) for r in range(world)] if bounds is None else [(int(bounds[r]), int(bounds[r + 1])) for r in range(world)]
    counts = [(c1 - c0) * I1 for c0, c1 in cols]
    cmax = max(counts)
    cplx = y_local.is_complex()
    src = torch.view_as_real(y_local) if cplx else y_local
    lead = tuple(y_local.shape[:-1])
    ax = len(lead)
    pad = torch.zeros(lead + (2 * cmax,) + tuple(src.shape[ax + 1:]), dtype=src.dtype, device=src.device)
    cnt = src.shape[ax] // 2
    pad.narrow(ax, 0, cnt).copy_(src.narrow(ax, 0, cnt))
    pad.narrow(ax, cmax, cnt).copy_(src.narrow(ax, cnt, cnt))
    out = torch.empty((world,) + tuple(pad.shape), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, pad.reshape((1,) + tuple(pad.shape)).contiguous(), group=group)
    parts = [out[r].narrow(ax, 0, counts[r]) for r in range(world)] + [out[r].narrow(ax, cmax, counts[r]) for r in reversed(range(world))]
    full = torch.cat(parts, dim=ax)
    return torch.view_as_complex(full.contiguous()) if cplx else full


class ShardedDasPlan:
    """``DasPlan`` for rank ``rank`` of ``world``: beamforms this rank's pixel slab and gathers the image.

    Two layouts.  Plain: contiguous slabs of the linear pixel index.  Mirror slabs (``mirror_slabs``; default: tried whenever the mode could
    apply -- ``'DAS'``, one image plane, an even number of columns -- and kept only if EVERY rank's plan accepted it, agreed with one
    ``all_reduce``): rank r takes columns ``[c0, c1)`` of the first half AND their mirror images, so that the lateral-mirror mode of the
    fused kernel (a pixel and its image share tap index and weights, ``csrc/tile_params.h``) survives the sharding.

    ``prefolded=True`` (a plan keyword): execute is handed FOLDED frames (``FoldedReplicator``), the rank runs no fold pass of its own.

    ``compute`` defaults to the HIP plan; the CPU tests inject the oracle there to exercise the sharding and the collective under ``gloo``
    without a GPU (``compute(xc, F, i_begin, i_count)``; with ``mirror_slabs=True`` it is called for slab A and for slab B).
    """

    def __init__(self, prob, rank: int, world: int, group=None, device=None, kernel: int = 0,
                 compute: Callable | None = None, mirror_slabs: bool | None = None, balance=None, balance_granule: int = 32, **plan_kw):
        """``balance`` (mirror slabs only): per-column (or per-block) cost of the first half of the image -- an array every rank passes identically, e.g. rank 0's
        :func:`measure_column_cost` after a broadcast -- or ``"measure"``: rank 0 measures it now and broadcasts it.  The ranks then take column ranges
        of equal COST (:func:`balanced_column_bounds`) instead of equal width -- in multiples of ``balance_granule`` columns (the fused kernel's tiles are 16-128
        columns wide and a partial column tile costs a whole one: see there)."""
        self.prob, self.rank, self.world, self.group = prob, rank, world, group
        self._compute = compute
        self.plan = None
        self.col_bounds = None
        I1, I2, I3 = prob.Isz
        can = world > 1 and prob.fun == "DAS" and I3 == 1 and I2 % 2 == 0 and I2 >= 2 and plan_kw.get("mirror", True)
        self.mirror_slabs = False
        if mirror_slabs is None:
            mirror_slabs = can and compute is None
        if mirror_slabs and can:
            if balance is not None:
                cost = self._agreed_cost(balance, device, kernel, plan_kw)
                if cost is not None:
                    self.col_bounds = balanced_column_bounds(expand_block_cost(cost, I2 // 2), world, max(1, int(balance_granule)))
            c0, c1 = mirror_slab_columns(I2, rank, world) if self.col_bounds is None else (self.col_bounds[rank], self.col_bounds[rank + 1])
            self.i_begin, self.i_count = c0 * I1, (c1 - c0) * I1
            ok = True
            if compute is None and self.i_count:
                from .das_spec import DasPlan
                from ._lib import QdasError
                failure = None
                try:
                    self.plan = DasPlan(prob, device=device, kernel=kernel, i_begin=self.i_begin, i_count=self.i_count, mirror_slab=True, **plan_kw)
                except QdasError as ex:                 # QDAS_EUNSUPPORTED (2): no lateral-mirror mode for this problem / this slab -> plain slabs.
                    ok = False                          # Anything else (out of memory, a bad device, an invalid argument) is a real error: it still
                    if ex.code != 2:                    # takes part in the agreement below -- the other ranks are waiting in it -- and is raised after.
                        failure = ex
                except Exception as ex:
                    ok, failure = False, ex
            else:
                failure = None
            ok = self._all_agree(ok, device)
            if failure is not None:
                raise failure
            if ok:
                self.mirror_slabs = True
                return
            if self.plan is not None:
                self.plan.close()
                self.plan = None
        self.i_begin, self.i_count = shard_range(prob.I, rank, world)
        if compute is None and self.i_count:
            from .das_spec import DasPlan
            self.plan = DasPlan(prob, device=device, kernel=kernel, i_begin=self.i_begin, i_count=self.i_count, **plan_kw)

    def _agreed_cost(self, balance, device, kernel, plan_kw):
        """the cost profile every rank uses: the caller's array as it is (the caller vouches that the ranks pass the same one), or rank 0's measurement, broadcast"""
        import numpy as np
        import torch
        import torch.distributed as dist
        if not isinstance(balance, str):
            return np.asarray(balance, dtype=np.float64).reshape(-1)
        if balance != "measure":
            raise ValueError("balance: an array of column / block costs, or 'measure'")
        nb = min(16, max(1, self.prob.Isz[1] // 2))
        t = torch.zeros(nb, dtype=torch.float64)
        if self.rank == 0:
            try:
                cost, _ = measure_column_cost(self.prob, device=device, nblocks=nb, kernel=kernel, **{k: v for k, v in plan_kw.items() if k != "prefolded"})
                t = torch.from_numpy(np.asarray(cost, dtype=np.float64))
            except Exception:                               # (no mirror mode for this problem, no device ...: zeros = "no profile", equal widths)
                t = torch.zeros(nb, dtype=torch.float64)
        if dist.is_initialized() and self.world > 1:
            on_gpu = dist.get_backend(self.group) == "nccl"
            tt = t.to((device if device is not None else "cuda") if on_gpu else "cpu")
            dist.broadcast(tt, src=(dist.get_global_rank(self.group, 0) if self.group is not None else 0), group=self.group)      # (rank 0 OF THE GROUP measured)
            t = tt.cpu()
        return None if float(t.sum()) <= 0 else t.numpy()

    def _all_agree(self, ok: bool, device) -> bool:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized() or self.world == 1:
            return ok
        on_gpu = dist.get_backend(self.group) == "nccl"
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=(device if device is not None else "cuda") if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()))

    @property
    def out_count(self) -> int:
        return 2 * self.i_count if self.mirror_slabs else self.i_count

    def execute_local(self, xc, F: int = 1, out=None):
        """this rank's share: ``(F, oM, oN, out_count)`` -- the slab, or ``[slab A | slab B]``"""
        import torch
        oN, oM = self.prob.osize
        if not self.i_count:                             # more ranks than pixels / columns
            return torch.zeros((F, oM, oN, 0), dtype=xc.dtype, device=xc.device)
        if self._compute is not None:
            if not self.mirror_slabs:
                return self._compute(xc, F, self.i_begin, self.i_count)
            ib = self.prob.I - self.i_begin - self.i_count
            return torch.cat([self._compute(xc, F, self.i_begin, self.i_count), self._compute(xc, F, ib, self.i_count)], dim=-1)
        return self.plan.execute_into(xc, out, F) if out is not None else self.plan.execute_colmajor(xc, F)

    def gather(self, y):
        if self.mirror_slabs:
            return gather_mirror_slabs(y, self.prob.Isz[0], self.prob.Isz[1], self.world, self.group, bounds=self.col_bounds)
        return gather_pixels(y, self.prob.I, self.world, self.group)

    def execute_colmajor(self, xc, F: int = 1):
        """``(F, oM, oN, I)`` on every rank (see :meth:`DasPlan.execute_colmajor`)."""
        return self.gather(self.execute_local(xc, F))

    def close(self):
        if self.plan is not None:
            self.plan.close()


# ------------------------------------------------------------------------------------------
# Replicating a reciprocal acquisition FOLDED.  Pixel slabs replicate the channel data; for a full-synthetic-aperture frame that is the larger part of a
# multi-GPU step (1.48 GB at C3 over xGMI against ~2-4 ms of kernel per rank).  A reciprocal frame can travel folded (csrc/fold.hip: xs[:, n, m] =
# x[:, n, m] + x[:, m, n] for n <= m -- exact by the linearity of the interpolators): the acquisition rank folds ONCE (qdas_fold), packs the upper
# triangle (N (N + 1) / 2 of the N x N traces: half the bytes), ONE broadcast moves it, every rank unpacks it into its folded-frame buffer and hands
# it to a QDAS_PLAN_PREFOLDED plan (ShardedDasPlan(..., prefolded=True)) -- which also takes the per-rank fold pass, the Amdahl term of this layout
# (DESIGN.md section 7), out of the ranks.  The reference has no counterpart (one gpuDevice per process, README.md:232).
# ------------------------------------------------------------------------------------------
def triangle_rows(N: int, device=None):
    """row indices ``m * N + n`` (``n <= m``) of the upper triangle in a column-major frame ``(M, N, T)`` viewed as ``(M * N, T)``: the traces the fold writes"""
    import torch
    m = torch.arange(N, device=device).repeat_interleave(torch.arange(1, N + 1, device=device))
    n = torch.cat([torch.arange(k + 1, device=device) for k in range(N)]) if N else torch.zeros(0, dtype=torch.long, device=device)
    return m * N + n


def fold_frame(xc, out=None, wtab=None):
    """``qdas_fold`` of one column-major frame ``xc`` ``(M, N, T)`` (complex64 or complex32, ``M == N``, on a HIP device) into the complex64 folded frame
    ``out`` (same shape; allocated zero-filled when not given: only the upper triangle ``n <= m`` is written)"""
    import ctypes as C
    import torch
    from . import _lib
    M, N, T = xc.shape
    if M != N:
        raise ValueError("a reciprocal frame has as many transmits as receivers")
    if out is None:
        out = torch.zeros((M, N, T), dtype=torch.complex64, device=xc.device)
    d = _lib.FoldDesc(T, N, 0, 0, 1 if xc.dtype == torch.complex64 else 2, xc.device.index if xc.device.index is not None else -1,
                      None if wtab is None else C.c_void_p(wtab.data_ptr()))
    with torch.cuda.device(xc.device):
        _lib.check(_lib.lib().qdas_fold(C.byref(d), C.c_void_p(xc.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(xc.device).cuda_stream)))
    return out


def pack_triangle(xs, rows=None):
    """the ``N (N + 1) / 2`` folded traces of ``xs`` ``(N, N, T)`` as one contiguous ``(N (N + 1) / 2, T)`` tensor (what travels)"""
    N, _, T = xs.shape
    rows = triangle_rows(N, xs.device) if rows is None else rows
    return xs.reshape(N * N, T).index_select(0, rows)


def unpack_triangle(packed, xs, rows=None):
    """the inverse of :func:`pack_triangle`: ``packed`` into the upper triangle of the folded-frame buffer ``xs`` ``(N, N, T)`` (the rest of it is never read)"""
    N, _, T = xs.shape
    rows = triangle_rows(N, xs.device) if rows is None else rows
    xs.reshape(N * N, T).index_copy_(0, rows, packed)
    return xs


class FoldedReplicator:
    """A stream of reciprocal frames from ONE acquisition rank to every rank of the group, folded: ``src`` calls :meth:`send` with a frame (any rank may
    pass ``None``), every rank then owns the folded frame ``(N, N, T)`` complex64 for its ``ShardedDasPlan(..., prefolded=True)``.  Two buffers:
    ``send(..., async_op=True)`` returns the work handle, so the transfer of frame ``f + 1`` overlaps the beamforming of frame ``f``.

    ``fold`` replaces ``qdas_fold`` in the CPU tests (``fold(xc) -> xs``) to exercise packing and the collective under ``gloo`` without a GPU."""

    def __init__(self, N: int, T: int, device, src: int = 0, group=None, fold: Callable | None = None, nbuf: int = 2):
        import torch
        self.N, self.T, self.src, self.group, self._fold = N, T, src, group, fold
        self.rows = triangle_rows(N, device)
        self.packed = [torch.zeros((N * (N + 1) // 2, T), dtype=torch.complex64, device=device) for _ in range(nbuf)]
        self.frames = [torch.zeros((N, N, T), dtype=torch.complex64, device=device) for _ in range(nbuf)]
        self._k = 0

    @property
    def bytes_per_frame(self) -> int:
        return self.packed[0].numel() * 8

    def send(self, xc, rank: int, async_op: bool = False):
        """start replicating one frame; returns ``(slot, work)`` -- ``work`` is None for a blocking call or a one-rank group"""
        import torch
        import torch.distributed as dist
        slot = self._k % len(self.packed)
        self._k += 1
        if rank == self.src:
            xs = self._fold(xc) if self._fold is not None else fold_frame(xc, self.frames[slot])
            if xs is not self.frames[slot]:
                self.frames[slot].copy_(xs)
            torch.index_select(self.frames[slot].reshape(self.N * self.N, self.T), 0, self.rows, out=self.packed[slot])
        work = None
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            work = dist.broadcast(torch.view_as_real(self.packed[slot]), self.src, group=self.group, async_op=async_op)
        return slot, (work if async_op else None)

    def receive(self, slot: int, rank: int, work=None):
        """the folded frame of ``slot`` on this rank (waits for ``work``; the acquisition rank already holds it)"""
        if work is not None:
            work.wait()
        if rank != self.src:
            unpack_triangle(self.packed[slot], self.frames[slot], self.rows)
        return self.frames[slot]


# ------------------------------------------------------------------------------------------
# Alternative layout (SURVEY.md section 8e "alternative"): shard the TRANSMITS.  Every rank holds only its own slice
# x[:, :, m-slab] of the channel data (1/G of it: for acquisitions that do not fit -- or should not be replicated --
# on every GPU), beamforms ALL pixels over that sub-aperture and the partial images are summed with one all_reduce
# (8 MiB at C3).  Works for the modes that sum over transmits ('DAS', 'SYN').  A sub-aperture of a full-synthetic-
# aperture acquisition is no longer reciprocal, so the tiled kernel's reciprocal mode does not engage here.
# ------------------------------------------------------------------------------------------
def slice_transmits(Pv, Nv, t0, apods, M: int, rank: int, world: int):
    """The rank's transmit slab ``[M*rank/world, M*(rank+1)/world)`` of every argument that has a transmit dimension:
    ``Pv`` / ``Nv`` (3 x M; single-column arrays broadcast and are kept), ``t0`` (scalar or ``1 x 1 x M``), apodization arrays
    (dimension 5 of ``I1 x I2 x I3 x N x M`` when it has size M).  Returns ``(Pv, Nv, t0, apods, m_begin, m_count)``."""
    import numpy as np
    b, c = shard_range(M, rank, world)
    sl = slice(b, b + c)
    cut = lambda P: P if np.asarray(P).reshape(np.asarray(P).shape[0], -1).shape[1] == 1 else np.asarray(P).reshape(np.asarray(P).shape[0], -1)[:, sl]
    t0a = np.asarray(t0)
    t0s = t0 if t0a.size == 1 else t0a.reshape(-1)[sl].reshape(1, 1, -1)
    out = []
    for a in apods:
        a = np.asarray(a)
        a5 = a.reshape(a.shape + (1,) * (5 - a.ndim))
        out.append(a5[..., sl] if a5.shape[4] == M and M > 1 else a5)
    return cut(Pv), cut(Nv), t0s, out, b, c


def allreduce_image(y, group=None):
    """Sum the ranks' partial images (complex tensors travel as (re, im) pairs of the real type)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return y
    buf = torch.view_as_real(y.contiguous()) if y.is_complex() else y.contiguous()
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return torch.view_as_complex(buf) if y.is_complex() else buf


def das_spec_tx_sharded(fun, Pi, Pr, Pv, Nv, x_local, t0, fs, c, *varargin, rank: int, world: int, M: int, group=None, compute: Callable | None = None):
    """``das_spec`` over a transmit-sharded acquisition: ``x_local`` is THIS rank's ``T x N x M_r`` slice; the result (the full
    image, summed over all transmits) is returned on every rank.  ``compute(fun, Pi, Pr, Pv_r, Nv_r, x_local, t0_r, fs, c, *opts)``
    replaces the device path in the CPU tests."""
    if fun not in ("DAS", "SYN"):
        raise ValueError("transmit sharding needs a mode that sums over transmits ('DAS' | 'SYN')")
    opts, apods, k = [], [], 0
    va = list(varargin)
    while k < len(va):
        if not isinstance(va[k], str):
            opts.append(va[k]); k += 1
        elif va[k] == "apod":
            apods.append(va[k + 1]); k += 2
        elif va[k] in ("input-precision", "device", "interp", "modulation", "transpose", "rx-apod"):
            opts += va[k:k + 2]; k += 2
        else:
            opts.append(va[k]); k += 1
    Pv_r, Nv_r, t0_r, ap_r, _, cnt = slice_transmits(Pv, Nv, t0, apods, M, rank, world)
    for a in ap_r:
        opts += ["apod", a]
    if cnt == 0:
        raise ValueError("transmit sharding needs at least one transmit per rank")
    if compute is not None:
        y = compute(fun, Pi, Pr, Pv_r, Nv_r, x_local, t0_r, fs, c, *opts)
    else:
        from .das_spec import das_spec
        y = das_spec(fun, Pi, Pr, Pv_r, Nv_r, x_local, t0_r, fs, c, *opts)
    return allreduce_image(y, group)
