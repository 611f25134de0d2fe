"""Deterministic synthetic inputs for the benchmark configs.

Every (gene, cell) value is a pure function of (seed, g, c) -- a splitmix64
hash -> Box-Muller -- so any block can be generated independently on the host
(NumPy) or directly in HBM (torch, integer ops wrap the same way), and a rank
of a cell-sharded run generates exactly its own columns.

Shape of the data: a log2(x+1)-scale matrix (the input of step 8) with
10 000 genes in 22 chromosome blocks sized like the reference's bundled gene
position file, the first 10 % of cells being reference cells in two groups, the
rest four tumour clones carrying arm-level gains (x1.5) and losses (x0.5).
"""
from __future__ import annotations

import numpy as np

SEED = 20250924
# genes per chr1..22 of inst/extdata/gencode_downsampled.EXAMPLE_ONLY_DONT_REUSE.txt
# (tests/golden/gencode_genes_per_chr.txt), chr1 += 61 so that the total is 10 000
CHR_SIZES_10K = (1072, 708, 607, 350, 480, 533, 514, 354, 410, 428, 611, 556, 187, 351, 318, 450, 628, 148, 656,
                 283, 108, 248)
# realistic i6 emission parameters: data/mcmc_obj.rda @mu, 1/sqrt(@sig)  
I6_MEANS = (0.41234766, 0.84075773, 1.01693983, 1.12238786, 1.23842619, 1.44298781)
I6_SDS = (0.02889, 0.16455, 0.10555, 0.19057, 0.24409, 0.29007)

_M1, _M2, _GOLD = 0xBF58476D1CE4E5B9, 0x94D049BB133111EB, 0x9E3779B97F4A7C15
_CG, _CC = 0xD1B54A32D192ED03, 0x8CB92BA72F3D8DD7
_MASK = (1 << 64) - 1


def chr_layout(G=10000):
    """Chromosome sizes for G genes: the 10k layout scaled, every chr >= 1 gene."""
    if G == 10000:
        sizes = np.array(CHR_SIZES_10K, dtype=np.int64)
    else:
        base = np.array(CHR_SIZES_10K, dtype=np.float64)
        sizes = np.maximum(1, np.floor(base * G / base.sum())).astype(np.int64)
        sizes[0] += G - sizes.sum()
        if sizes[0] < 1:
            raise ValueError("G too small for 22 chromosomes")
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)


def groups(C, ref_frac=0.10, n_ref_groups=2, n_clones=4):
    """(ref_groups, obs_groups): lists of 0-based global cell index arrays."""
    n_ref = max(n_ref_groups, int(round(C * ref_frac)))
    bounds = np.linspace(0, n_ref, n_ref_groups + 1).astype(np.int64)
    refs = [np.arange(bounds[i], bounds[i + 1], dtype=np.int32) for i in range(n_ref_groups)]
    obs_all = np.arange(n_ref, C, dtype=np.int32)
    obs = [obs_all[(obs_all % n_clones) == q] for q in range(n_clones)]
    return refs, obs


def subclusters(C, size=500, ref_frac=0.10, n_ref_groups=2, n_clones=4):
    """BASELINE configs 4 / 5: consecutive blocks of `size` members within each annotation group.
    Returns (subclusters as global cell index arrays -- reference groups first --, is_ref flag per subcluster, cut
    candidates: the cell offsets at which no subcluster is split, for sharded.align_to_groups)."""
    refs, obs = groups(C, ref_frac, n_ref_groups, n_clones)
    subs, is_ref = [], []
    for flag, grp in ((True, refs), (False, obs)):
        for g in grp:
            for s in range(0, len(g), size):
                subs.append(g[s:s + size])
                is_ref.append(flag)
    n_ref = int(sum(len(r) for r in refs))
    cuts = {0, C, n_ref}
    for r in refs:
        cuts.update(int(r[0]) + k for k in range(0, len(r), size))
    cuts.update(range(n_ref, C, size * n_clones))          # the clones interleave: a block of every clone spans size * n_clones columns
    return subs, is_ref, sorted(cuts)


def _cnv_factor_table(chr_start, n_clones=4):
    """k[clone+1, chr]: clone q gains on chr (1+q),(7+q), losses on chr (10+q),(17+q) (1-based); row 0 = reference."""
    n_chr = len(chr_start) - 1
    tab = np.ones((n_clones + 1, n_chr), dtype=np.float64)
    for q in range(n_clones):
        for c, f in ((1 + q, 1.5), (7 + q, 1.5), (10 + q, 0.5), (17 + q, 0.5)):
            if c - 1 < n_chr:
                tab[q + 1, c - 1] = f
    return tab


# ------------------------------------------------------------------ NumPy
def _mix_np(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(_M1)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(_M2)
    return z ^ (z >> np.uint64(31))


def _uniforms_np(seed, g, c):
    with np.errstate(over="ignore"):
        s = np.uint64(seed) + g.astype(np.uint64) * np.uint64(_CG) + c.astype(np.uint64) * np.uint64(_CC)
        h1 = _mix_np(s + np.uint64(_GOLD))
        h2 = _mix_np(s + np.uint64((2 * _GOLD) & _MASK))
    u1 = ((h1 >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    u2 = ((h2 >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    return u1, u2


def _normal_np(seed, g, c):
    u1, u2 = _uniforms_np(seed, g, c)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def make_matrix_np(G, C, cell_offset=0, C_total=None, seed=SEED, n_clones=4, ref_frac=0.10):
    """(G, C) float64 Fortran-ordered block of cells [cell_offset, cell_offset+C) of a C_total-cell matrix."""
    C_total = C if C_total is None else C_total
    chr_start = chr_layout(G)
    n_ref = max(2, int(round(C_total * ref_frac)))
    g = np.arange(G, dtype=np.int64)[:, None]
    c = (np.arange(C, dtype=np.int64) + cell_offset)[None, :]
    z = _normal_np(seed, np.broadcast_to(g, (G, C)), np.broadcast_to(c, (G, C)))
    zg = _normal_np(seed, g[:, 0], np.full(G, 0xFFFFFFFF, dtype=np.int64))
    m_g = np.exp(1.0 + 1.5 * zg)[:, None]
    tab = _cnv_factor_table(chr_start, n_clones)
    chr_of_gene = np.repeat(np.arange(len(chr_start) - 1), np.diff(chr_start))
    clone = np.where(c[0] < n_ref, 0, 1 + (c[0] % n_clones))
    k = tab[clone][:, chr_of_gene].T
    x = np.log2(1.0 + m_g * k * np.exp(0.5 * z))
    return np.asfortranarray(x), chr_start


# ------------------------------------------------------------------ torch (device)
def _mix_t(z):
    import torch
    def lsr(x, n):
        return (x >> n) & ((1 << (64 - n)) - 1)
    m1 = _M1 - (1 << 64)
    m2 = _M2 - (1 << 64)
    z = (z ^ lsr(z, 30)) * m1
    z = (z ^ lsr(z, 27)) * m2
    return z ^ lsr(z, 31)


def _normal_t(seed, g, c):
    import torch
    def s64(v):
        v &= _MASK
        return v - (1 << 64) if v >= (1 << 63) else v
    s = g * s64(_CG) + c * s64(_CC) + s64(seed)
    h1 = _mix_t(s + s64(_GOLD))
    h2 = _mix_t(s + s64(2 * _GOLD))
    def u(h):
        return (((h >> 11) & ((1 << 53) - 1)).to(torch.float64) + 0.5) * (1.0 / 9007199254740992.0)
    return torch.sqrt(-2.0 * torch.log(u(h1))) * torch.cos(2.0 * torch.pi * u(h2))


def make_matrix_torch(G, C, device, cell_offset=0, C_total=None, seed=SEED, n_clones=4, ref_frac=0.10,
                      chunk_cells=8192, cell_stride=1):
    """(C, G) contiguous float64 CUDA tensor (cell-major) + chr_start; same values as make_matrix_np up to
    libm-vs-device rounding of log/exp/cos.  Local cell i is global cell cell_offset + cell_stride * i
    (a contiguous block: offset c0, stride 1; the round-robin deal of sharded.cyclic_cells: offset rank, stride world)."""
    import torch
    C_total = C if C_total is None else C_total
    chr_start = chr_layout(G)
    n_ref = max(2, int(round(C_total * ref_frac)))
    out = torch.empty((C, G), dtype=torch.float64, device=device)
    g = torch.arange(G, dtype=torch.int64, device=device)
    zg = _normal_t(seed, g, torch.full((G,), 0xFFFFFFFF, dtype=torch.int64, device=device))
    m_g = torch.exp(1.0 + 1.5 * zg)[None, :]
    tab = torch.as_tensor(_cnv_factor_table(chr_start, n_clones), device=device)
    chr_of_gene = torch.as_tensor(np.repeat(np.arange(len(chr_start) - 1), np.diff(chr_start)), device=device)
    for c0 in range(0, C, chunk_cells):
        c1 = min(C, c0 + chunk_cells)
        c = torch.arange(c0, c1, dtype=torch.int64, device=device) * cell_stride + cell_offset
        z = _normal_t(seed, g[None, :], c[:, None])
        clone = torch.where(c < n_ref, torch.zeros_like(c), 1 + (c % n_clones))
        k = tab[clone][:, chr_of_gene]
        out[c0:c1] = torch.log2(1.0 + m_g * k * torch.exp(0.5 * z))
    return out, chr_start


def hmm_params_i6(t=1e-6):
    """(means, shared sd = median(sds), logPi col-major, logDelta) for the benchmark's i6 HMM."""
    Pi = np.full((6, 6), t)
    np.fill_diagonal(Pi, 1 - 5 * t)
    delta = np.array([t, t, 1 - 5 * t, t, t, t])
    sds = np.sort(np.array(I6_SDS))
    return np.array(I6_MEANS), float((sds[2] + sds[3]) * 0.5), np.log(Pi), np.log(delta)
