"""Cell-sharded multi-GPU execution: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Cells are independent in every stage of the hot path, so each
rank owns its share of the cells -- a contiguous block, a round-robin deal
(`cyclic_cells`) or whole groups -- and there is NO data-path collective.
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

    def __init__(self, engine, process_group=None, world_size=None, always_reduce=False):
        self.engine = engine
        self.pg = process_group
        self.world = world_size
        self.always_reduce = always_reduce      # issue the collective on a communicator of one rank too (bench.py's RCCL smoke)

    def _all_reduce(self, buf):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.pg) > 1 or self.always_reduce):
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)

    def run(self, x_local, out=None, want_pre_denoise=False, pre=None):
        for r in range(self.engine.num_rounds):
            buf = self.engine.round_partial(r, x_local)
            self._all_reduce(buf)
            self.engine.round_finish(r)
        if pre is not None:                      # a preallocated HMM-input matrix (engines without the keyword allocate their own)
            return self.engine.apply(x_local, out=out, want_pre_denoise=True, pre=pre)
        return self.engine.apply(x_local, out=out, want_pre_denoise=want_pre_denoise)


# ---------------------------------------------------------------------------------------------------------------------
# Group-level HMM (BASELINE config 4: i3 at subcluster level) and the 2-D median filter (config 5) on cell shards.
#
# Groups are independent in the group HMM (R/inferCNV_HMM.R:371, 529-533; R/inferCNV_i3HMM.R:249-308) and tiles in the
# median filter (R/noise_reduction.R:57-86), so the partition puts every group / tile WHOLE on one rank: no halo, no
# cross-rank mean.  RCCL call sequence per step (all latency-bound, f64, sum):
#   group HMM, i3:  all-reduce of 2 doubles {sum x, n} over the reference values     -> mu
#                   all-reduce of 2 doubles {sum (x - mu)^2, n}                      -> sigma, delta = |qnorm(p, 0, sigma)|
#                   then group means -> Viterbi -> broadcast, rank-local
#   group HMM, i6:  none (the per-group sd comes from the hspike fit, a host-side input)
#   median filter:  none
def assign_groups(group_sizes, world: int):
    """Whole groups to ranks: longest group first onto the least loaded rank (ties: lowest rank) -- deterministic.
    Returns one ascending list of group ids per rank."""
    sizes = [int(s) for s in group_sizes]
    load = [0] * world
    out = [[] for _ in range(world)]
    for gid in sorted(range(len(sizes)), key=lambda i: (-sizes[i], i)):
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(gid)
        load[r] += sizes[gid]
    return [sorted(v) for v in out]


def gather_groups(groups, group_ids):
    """The cells a rank holds when it owns `group_ids`: (global cell ids in storage order, the groups as LOCAL index lists).
    A cell that sits in several owned groups is stored once."""
    cells, pos = [], {}
    local = []
    for gid in group_ids:
        idx = []
        for c in np.asarray(groups[gid], dtype=np.int64):
            c = int(c)
            if c not in pos:
                pos[c] = len(cells)
                cells.append(c)
            idx.append(pos[c])
        local.append(np.asarray(idx, dtype=np.int32))
    return np.asarray(cells, dtype=np.int64), local


def _all_reduce_pair(a, b, pg=None, device=None):
    """all-reduce(sum) of two doubles; a no-op without an initialised process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(pg) > 1):
        return float(a), float(b)
    buf = torch.tensor([a, b], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=pg)
    v = buf.cpu().tolist()
    return v[0], v[1]


def sharded_mean_sd(moments_partial, pg=None, device=None):
    """mean and sd over ALL reference values of a cell-sharded matrix (R/inferCNV_i3HMM.R:38-52: mean(ref), sd(ref)):
    `moments_partial(phase, mean)` returns this rank's {sum, n} (phase 0) or {sum (x - mean)^2, n} (phase 1) --
    icnv_cells_moments_partial_dev; two all-reduces of two doubles."""
    s, n = _all_reduce_pair(*moments_partial(0, 0.0), pg=pg, device=device)
    if not n > 1:
        raise ValueError("fewer than two reference values")
    mean = s / n
    ss, n2 = _all_reduce_pair(*moments_partial(1, mean), pg=pg, device=device)
    return mean, float(np.sqrt(ss / (n2 - 1.0)))


class DeviceGroupEngine:
    """The rank-local pieces of the group HMM / median filter on libicnv_hip.so (device-resident tensors)."""

    def moments_partial(self, x_local, ref_local, phase, mean):
        from . import device
        return device.cells_moments_partial(x_local, ref_local, phase, mean)

    def viterbi_groups(self, x_local, chr_start, groups_local, means, sds, logPi, logDelta):
        from . import device
        return device.viterbi_groups(x_local, chr_start, groups_local, means, sds, logPi, logDelta)[0]

    def median_filter(self, x_local, chr_start, tiles_local, window_size):
        from . import device
        return device.median_filter(x_local, chr_start, tiles_local, window_size)


class ShardedGroupHMM:
    """predict_CNV_via_HMM_on_tumor_subclusters / _whole_tumor_samples and their i3 forms on a cell-sharded matrix whose
    groups are whole on their ranks (assign_groups + gather_groups, or contiguous blocks cut by align_to_groups)."""

    def __init__(self, engine=None, process_group=None, device=None):
        self.engine = engine or DeviceGroupEngine()
        self.pg = process_group
        self.device = device

    def i3_params(self, x_local, ref_local, i3_p_val=0.05):
        """(mu, sigma, delta) of .i3HMM_get_sd_trend_by_num_cells_fit (R/inferCNV_i3HMM.R:17-80, 435-445)."""
        import statistics
        mu, sigma = sharded_mean_sd(lambda ph, m: self.engine.moments_partial(x_local, ref_local, ph, m), self.pg, self.device)
        return mu, sigma, abs(statistics.NormalDist(0.0, sigma).inv_cdf(i3_p_val))

    def _plan_for(self, x_local, chr_start, groups_local, ref_local):
        """The device plan of this (matrix shape, layout, groups, reference cells), built on first use and kept: the group
        structure is uploaded once, not per step.  None when the engine is not the device engine or the plan does not take
        the groups (a cell in two groups, more than 8192 sequences): run_i3 then takes the call-by-call path."""
        if not isinstance(self.engine, DeviceGroupEngine):
            return None
        key = (tuple(x_local.shape), np.asarray(chr_start).tobytes(), tuple(np.asarray(g).tobytes() for g in groups_local),
               np.asarray(ref_local).tobytes())
        if getattr(self, "_plan_key", None) != key:
            from . import device
            self._plan, self._plan_key = None, key
            try:
                self._plan = device.GroupHMMPlan(x_local.shape[1], x_local.shape[0], chr_start, groups_local, ref_local)
            except RuntimeError as e:                # ICNV_ERR_UNSUPPORTED (3): the call-by-call path serves these groups
                if getattr(e, "code", None) != 3:
                    raise
                self._plan = None
        return self._plan

    def run_i3(self, x_local, chr_start, groups_local, ref_local, t=1e-6, i3_p_val=0.05, ks_delta=None):
        """i3HMM_predict_CNV_via_HMM_on_tumor_subclusters (R/inferCNV_i3HMM.R:249-308): states of this rank's cells.
        The state means sit at mu +- delta: the Z-based delta of use_KS = FALSE, or `ks_delta` -- the KS-based one of the
        reference's default use_KS = TRUE, a function of (sigma, p, number of reference cells, RNG state) alone
        (hmm.get_HoneyBADGER_setGexpDev), so every rank computes the same value from the all-reduced sigma."""
        Pi = np.full((3, 3), t)
        np.fill_diagonal(Pi, 1.0 - 5.0 * t)               # the reference's 1 - 5t diagonal with three states (:108-112)
        d0 = np.array([t, 1.0 - 5.0 * t, t])
        plan = None if callable(ks_delta) else self._plan_for(x_local, chr_start, groups_local, ref_local)
        if plan is not None:
            # device-resident parameters (round 6): one pass for the group means AND the reference moments, the all-reduce of three
            # doubles on the library's own buffer, mu / sigma / delta derived on the device -- no host round trip in the step
            import statistics
            import torch.distributed as dist
            m3 = plan.i3_partial(x_local)
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.pg) > 1:
                dist.all_reduce(m3, op=dist.ReduceOp.SUM, group=self.pg)
            z = abs(statistics.NormalDist().inv_cdf(i3_p_val))
            return plan.i3_finish(np.log(Pi), np.log(d0), z, None if ks_delta is None else float(ks_delta), device=x_local.device)
        mu, sigma, delta = self.i3_params(x_local, ref_local, i3_p_val)
        if ks_delta is not None:
            delta = float(ks_delta(sigma)) if callable(ks_delta) else float(ks_delta)
        means = np.array([mu - delta, mu, mu + delta])
        return self.engine.viterbi_groups(x_local, chr_start, groups_local, means, [sigma] * len(groups_local),
                                          np.log(Pi), np.log(d0))

    def run_i6(self, x_local, chr_start, groups_local, means, sd_per_group, logPi, logDelta):
        """predict_CNV_via_HMM_on_tumor_subclusters (R/inferCNV_HMM.R:345-408): no exchange at all."""
        return self.engine.viterbi_groups(x_local, chr_start, groups_local, means, sd_per_group, logPi, logDelta)


class ShardedMedianFilter:
    """apply_median_filtering (R/noise_reduction.R:43-113) on a cell-sharded matrix whose tiles are whole on their ranks:
    rank-local, no collective."""

    def __init__(self, engine=None):
        self.engine = engine or DeviceGroupEngine()

    def run(self, x_local, chr_start, tiles_local, window_size=7):
        return self.engine.median_filter(x_local, chr_start, tiles_local, window_size)


class ShardedIngest:
    """Steps 2-4 of run() from integer counts on cell shards.  Exchange steps:
        all-reduce(sum) of [G gene sums | G numbers of expressing cells]   -> the filter decision, identical on every rank
        all-gather of the ranks' column sums over the kept genes          -> median(colSums) = the normalisation factor
    `engine` has gene_stats(counts) -> 2G vector, col_sums(counts, keep) -> C_local vector and apply(counts, keep, col_sums,
    factor) -> matrix; the default one runs on libicnv_hip.so (device.ingest_*)."""

    def __init__(self, engine=None, process_group=None):
        self.engine = engine
        self.pg = process_group

    def run(self, counts_local, C_total, min_mean_expr_cutoff=None, min_cells_per_gene=0, normalize_factor=None):
        import torch
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.pg) > 1
        eng = self.engine
        if eng is None:
            from . import device as eng_mod
            class _Dev:
                gene_stats = staticmethod(eng_mod.ingest_gene_stats)
                col_sums = staticmethod(eng_mod.ingest_col_sums)
                apply = staticmethod(eng_mod.ingest_apply)
                select = staticmethod(eng_mod.ingest_select)
            eng = _Dev
        stats = eng.gene_stats(counts_local)
        if multi:
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.pg)
        stats_h = stats.cpu().numpy() if hasattr(stats, "cpu") else np.asarray(stats)
        G = stats_h.size // 2
        keep = eng.select(stats_h, G, C_total, min_mean_expr_cutoff, min_cells_per_gene)
        cs = eng.col_sums(counts_local, keep)
        factor = normalize_factor
        if factor is None:
            if multi:
                world = dist.get_world_size(self.pg)
                n_local = torch.tensor([cs.numel()], dtype=torch.int64, device=cs.device)
                sizes = [torch.zeros_like(n_local) for _ in range(world)]
                dist.all_gather(sizes, n_local, group=self.pg)
                nmax = int(max(int(v.item()) for v in sizes))
                pad = torch.zeros(nmax, dtype=cs.dtype, device=cs.device)
                pad[: cs.numel()] = cs
                parts = [torch.zeros_like(pad) for _ in range(world)]
                dist.all_gather(parts, pad, group=self.pg)
                allcs = np.concatenate([p[: int(n.item())].cpu().numpy() for p, n in zip(parts, sizes)])
            else:
                allcs = cs.cpu().numpy() if hasattr(cs, "cpu") else np.asarray(cs)
            factor = float(np.median(allcs))              # stats::median: mean of the two middle values for an even count
        return eng.apply(counts_local, keep, cs, factor), keep, factor
