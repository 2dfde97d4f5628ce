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


class ShardedDasPlan:
    """``DasPlan`` for rank ``rank`` of ``world``: beamforms this rank's pixel slab and gathers the image.

    ``compute`` defaults to the HIP plan; the CPU tests inject the oracle there to exercise the sharding and
    the collective under ``gloo`` without a GPU.
    """

    def __init__(self, prob, rank: int, world: int, group=None, device=None, kernel: int = 0,
                 compute: Callable | None = None):
        self.prob, self.rank, self.world, self.group = prob, rank, world, group
        self.i_begin, self.i_count = shard_range(prob.I, rank, world)
        self._compute = compute
        self.plan = None
        if compute is None:
            from .das_spec import DasPlan
            self.plan = DasPlan(prob, device=device, kernel=kernel, i_begin=self.i_begin, i_count=self.i_count)

    def execute_colmajor(self, xc, F: int = 1):
        """``(F, oM, oN, I)`` on every rank (see :meth:`DasPlan.execute_colmajor`)."""
        if self.i_count:
            y = self.plan.execute_colmajor(xc, F) if self._compute is None else self._compute(xc, F, self.i_begin, self.i_count)
        else:  # more ranks than pixels
            import torch
            oN, oM = self.prob.osize
            y = torch.zeros((F, oM, oN, 0), dtype=xc.dtype, device=xc.device)
        return gather_pixels(y, self.prob.I, self.world, self.group)
