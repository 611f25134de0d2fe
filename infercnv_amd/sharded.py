"""Cell-sharded multi-GPU execution: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Cells are independent in every stage of the hot path, so each
rank owns a contiguous block of cells and there is NO data-path collective.
The only exchange is the reference-normal statistics of the smoothing chain:
one small all-reduce(sum) per reference round (steps 8, 12 and 22 each need
the statistics of the *previous* stages' output on the reference cells, so the
rounds are dependent):

    subtract rounds: [G * n_ref_groups gene sums | n_ref_groups counts]  (160 KB at G=10k, 2 groups)
    denoise round:   [sum x, sum_c sd_c, n_ref_cells, n_ref_values]       (32 B)

The per-rank engine is `device.ChainPlan` (libicnv_hip.so); anything with the
same four methods can be plugged in (the gloo tests use an oracle-backed one).
"""
from __future__ import annotations

import numpy as np


def shard_bounds(C_total: int, world: int, rank: int):
    """Contiguous block [c0, c1) of rank `rank`; the first C_total % world ranks get one more cell."""
    base, rem = divmod(int(C_total), int(world))
    c0 = rank * base + min(rank, rem)
    return c0, c0 + base + (1 if rank < rem else 0)


def localize_groups(groups, c0: int, c1: int):
    """Keep each group's members that live in [c0, c1), as LOCAL indices, order preserved."""
    out = []
    for g in groups:
        g = np.asarray(g, dtype=np.int64)
        m = g[(g >= c0) & (g < c1)] - c0
        out.append(m.astype(np.int32))
    return out


def cyclic_cells(C_total: int, world: int, rank: int):
    """Global indices of the cells rank `rank` holds when cells are dealt round-robin: rank, rank + world, ...
    Every annotation group -- the reference groups in particular -- is then spread evenly over the ranks, so the
    reference rounds cost every rank the same (a contiguous block partition of a matrix whose reference cells
    come first would leave them all on rank 0)."""
    return np.arange(rank, C_total, world, dtype=np.int64)


def localize_groups_cyclic(groups, rank: int, world: int):
    """Members of each group held by `rank` under the round-robin deal, as LOCAL indices, order preserved."""
    out = []
    for g in groups:
        g = np.asarray(g, dtype=np.int64)
        m = g[(g % world) == rank]
        out.append(((m - rank) // world).astype(np.int32))
    return out


def align_to_groups(C_total: int, world: int, group_boundaries):
    """Shard boundaries moved to the nearest group boundary so that every group
    (HMM subcluster / median-filter tile) lives on one GPU.
    `group_boundaries`: sorted cell offsets at which a cut is allowed."""
    gb = np.asarray(sorted(set(int(b) for b in group_boundaries) | {0, int(C_total)}), dtype=np.int64)
    cuts = [0]
    for r in range(1, world):
        ideal = shard_bounds(C_total, world, r)[0]
        cand = gb[np.argmin(np.abs(gb - ideal))]
        cuts.append(int(max(cand, cuts[-1])))
    cuts.append(int(C_total))
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


class ShardedChain:
    """Drives one rank's engine through the reference rounds with an all-reduce
    between `round_partial` and `round_finish`, then the fused apply pass."""

    def __init__(self, engine, process_group=None, world_size=None):
        self.engine = engine
        self.pg = process_group
        self.world = world_size

    def _all_reduce(self, buf):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.pg) > 1:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)

    def run(self, x_local, out=None, want_pre_denoise=False):
        for r in range(self.engine.num_rounds):
            buf = self.engine.round_partial(r, x_local)
            self._all_reduce(buf)
            self.engine.round_finish(r)
        return self.engine.apply(x_local, out=out, want_pre_denoise=want_pre_denoise)
