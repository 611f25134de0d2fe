"""GPU parity tests (-m gpu): the HIP path, called through the C ABI of
libicnv_hip.so, against the CPU oracle on identical seeded inputs, against the
reference's golden object, and -- at BASELINE.json's full size -- through
size-independent properties.

Tolerances (stated by BASELINE.json's north_star):
  * smoothing chain: 1e-5 relative.  We assert far tighter absolute bounds
    (1e-10 .. 1e-12) because everything is computed in fp64.
  * HMM state calls: bit-exact on identical inputs.
  * median filter: exact (order statistics + one average).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import oracle_c as oc  # noqa: E402
import oracle_np as onp  # noqa: E402
from parity_util import check_denoise_flips  # noqa: E402

RTOL_NORTH_STAR = 1e-5


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from infercnv_amd import device
    torch.cuda.set_device(0)
    device.init(0)
    return device


@pytest.fixture(autouse=True)
def _default_viterbi_mode(dev):
    """Every test starts from the library's default Viterbi mode (auto), whatever an earlier test left behind."""
    dev.viterbi_set_mode(0)
    yield
    dev.viterbi_set_mode(0)


def to_dev(x_gc):
    """(G, C) host matrix -> (C, G) contiguous CUDA tensor (same bytes as R's column-major G x C)."""
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x_gc, dtype=np.float64).T)).cuda()


def to_host(t_cg):
    return t_cg.cpu().numpy().T


def rel_err(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.fixture(scope="module")
def example(golden_dir):
    d = np.load(os.path.join(golden_dir, "infercnv_object_example.npz"))
    ex = {k: d[k] for k in d.files}
    ex["log"] = onp.log2xplus1(onp.normalize_counts_by_seq_depth(ex["count_data"]))
    ex["chr_start"] = oc.chr_starts_from_codes(ex["chr_codes"])
    return ex


# ------------------------------------------------------------------ smoothing chain
def test_chain_reproduces_reference_golden_object(dev, example):
    """HIP chain on the reference's own example: @count.data -> @expr.data."""
    out, pre = dev.smooth_chain(to_dev(example["log"]), example["chr_start"], [example["ref_normal"]],
                                want_pre_denoise=True)
    got = to_host(out)
    gold = example["expr_data"]
    ref_out, ref_pre, (mu, s) = oc.smooth_chain(example["log"], example["chr_start"], [example["ref_normal"]],
                                                want_pre_denoise=True)
    assert rel_err(to_host(pre), ref_pre) < RTOL_NORTH_STAR
    assert np.abs(to_host(pre) - ref_pre).max() < 1e-12
    # step 22 is a strict select: an element may differ only if its pre-denoise value sits on a bound; on the
    # reference's own object no element does
    check_denoise_flips(got, gold, ref_pre, mu, s, tol=1e-11, expect=0, label="golden object vs @expr.data")
    check_denoise_flips(got, ref_out, ref_pre, mu, s, tol=1e-12, expect=0, label="golden object vs oracle")


# the gene counts exercise every kernel geometry: 768 x 7 / 15 / 23 and 512 x 37 positions, even and odd G
@pytest.mark.parametrize("G,C,nref", [(10000, 160, 2), (4613, 70, 1), (9999, 67, 3), (301, 33, 2), (37, 9, 1),
                                      (3000, 40, 2), (15000, 48, 2), (14999, 21, 1), (16600, 36, 2), (16501, 19, 1)])
def test_chain_fused_vs_oracle(dev, G, C, nref):
    from infercnv_amd import synth
    x, cs = synth.make_matrix_np(G, C)
    rng = np.random.default_rng(G)
    perm = rng.permutation(C)
    n_ref = max(nref, C // 5)
    cuts = np.linspace(0, n_ref, nref + 1).astype(int)
    refs = [perm[cuts[i]:cuts[i + 1]].astype(np.int32) for i in range(nref)]   # unsorted, non-contiguous
    out, pre = dev.smooth_chain(to_dev(x), cs, refs, want_pre_denoise=True)
    ref_out, ref_pre, (mu, s) = oc.smooth_chain(x, cs, refs, want_pre_denoise=True)
    assert rel_err(to_host(pre), ref_pre) < RTOL_NORTH_STAR
    assert np.abs(to_host(pre) - ref_pre).max() < 1e-11
    check_denoise_flips(to_host(out), ref_out, ref_pre, mu, s, tol=1e-11, label=f"fused G={G}")


@pytest.mark.parametrize("sizes", [(1,), (2,), (3,), (4,), (5,), (3, 4), (1, 1, 5), (6, 1)])
def test_chain_tiny_gene_counts(dev, sizes):
    """One to eight genes: below four genes the one-gene-per-slot kernels serve the cell, from four on the pair layout (odd
    counts with its repeated element)."""
    rng = np.random.default_rng(sum(sizes) * 7 + len(sizes))
    G, C = int(sum(sizes)), 21
    cs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    x = np.abs(rng.normal(2.0, 1.0, size=(G, C)))
    refs = [np.arange(0, 4, dtype=np.int32), np.arange(4, 9, dtype=np.int32)]
    for window in (3, 101):
        out, pre = dev.smooth_chain(to_dev(x), cs, refs, window_length=window, want_pre_denoise=True)
        ref_out, ref_pre, (mu, s) = oc.smooth_chain(x, cs, refs, window_length=window, want_pre_denoise=True)
        assert np.abs(to_host(pre) - ref_pre).max() < 1e-11, (sizes, window)
        check_denoise_flips(to_host(out), ref_out, ref_pre, mu, s, tol=1e-11, label=f"tiny {sizes} w{window}")


# (genes, window, kernel variant the geometry rules of chain_kernels.hip pick): the run-time-window and the
# compile-time-window (101) forms of 1024 x 11 / five slots and 768 x 15 / seven slots, and 768 x 15 / eight slots
@pytest.mark.parametrize("G,window,variant", [(9000, 61, "w11"), (10060, 101, "w11t"), (10400, 41, "m15s"),
                                              (10300, 101, "m15t"), (10900, 21, "m15"),
                                              # round 5: chunk lengths 17 / 19 / 21 with fitted slot counts between the 10 000-gene
                                              # geometries and 768 x 23 (even G: (L - 1) / 2 gene-pair slots; odd G: one gene per slot)
                                              (11000, 101, "m17 / 8 slots"), (11001, 101, "m17 odd"), (12400, 101, "m19 / 9 slots"),
                                              (14000, 101, "m21 / 10 slots"), (14001, 61, "m21 odd"), (16000, 101, "m23 / 11 slots"),
                                              # odd gene counts in the pair layout (round 5): the slot of the last gene repeats gene
                                              # G - 2; 10 239 = every slot of 1024 x 5 pairs taken, the repeated element the only spare
                                              (10239, 21, "w11 odd, all slots"), (9939, 101, "w11t odd"), (10301, 101, "m15t odd")])
def test_chain_geometries_and_windows(dev, G, window, variant):
    """Every (threads x chunk length, slots, window form) variant of the fused kernel against the oracle: the full chain,
    the chain without denoise, and stage subsets that run the generic (run-time mask) kernels of the same geometry."""
    from infercnv_amd import synth
    C = 300                                                  # more cells than workgroups
    x, cs = synth.make_matrix_np(G, C)
    refs = [np.arange(0, 17, dtype=np.int32), np.arange(17, 40, dtype=np.int32)]
    xd = to_dev(x)
    for mask in (0x7F, 0x3F):
        out, pre = dev.smooth_chain(xd, cs, refs, window_length=window, stage_mask=mask, want_pre_denoise=True)
        want_out, want_pre, musd = oc.smooth_chain(x, cs, refs, window_length=window, stage_mask=mask, want_pre_denoise=True)
        if mask == 0x3F:
            want_pre = want_out
        assert np.abs(to_host(pre) - want_pre).max() < 1e-11, (variant, hex(mask))
        if mask == 0x3F:
            assert np.abs(to_host(out) - want_out).max() < 1e-11, (variant, hex(mask))
        else:
            check_denoise_flips(to_host(out), want_out, want_pre, *musd, tol=1e-11, label=f"{variant} {mask:#x}")
    got = to_host(dev.smooth_chain(xd, cs, refs, window_length=window, stage_mask=0x0C)[0])   # smooth + centre only
    want = oc.center_columns(oc.smooth_by_chromosome(x, cs, window), "median")
    assert np.abs(got - want).max() < 1e-11, variant


STAGES = {"st8": 0x01, "st9": 0x02, "st10": 0x04, "st11": 0x08, "st12": 0x10, "st14": 0x20, "st22": 0x40,
          "st11mean": 0x88}


@pytest.mark.parametrize("stage", list(STAGES))
def test_chain_each_stage_standalone(dev, stage):
    """Every R-level wrapper is a single-bit stage_mask call (run(up_to_step=), resume)."""
    from infercnv_amd import synth
    G, C = 3001, 48          # odd G: the pair layout with one repeated element (round 5; before: one gene per slot)
    x, cs = synth.make_matrix_np(G, C)
    x = x - 2.0
    refs = [np.arange(0, 5, dtype=np.int32), np.arange(5, 9, dtype=np.int32)]
    mask = STAGES[stage]
    out, _ = dev.smooth_chain(to_dev(x), cs, refs, window_length=101, max_thresh=1.25, stage_mask=mask)
    got = to_host(out)
    if stage in ("st8", "st12"):
        want = oc.subtract_ref_expr_from_obs(x, refs)
    elif stage == "st9":
        want = oc.apply_max_threshold_bounds(x, 1.25)
    elif stage == "st10":
        want = oc.smooth_by_chromosome(x, cs, 101)
    elif stage == "st11":
        want = oc.center_columns(x, "median")
    elif stage == "st11mean":
        want = oc.center_columns(x, "mean")
    elif stage == "st14":
        want = oc.invert_log2(x)
    else:
        mu, s = oc.denoise_params(x, np.concatenate(refs), 1.5)
        want = oc.denoise_apply(x, mu, s)
    assert np.abs(got - want).max() < 1e-11 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("mask", [0x7F, 0x3F, 0x0F, 0x70, 0x30, 0x20, 0x01, 0x00])
def test_chain_many_cells_per_workgroup(dev, mask):
    """More cells than workgroups: every workgroup streams several cells through its software pipeline (prefetch of the
    next cell, stores of the previous one) -- for the full chain and for the stage subsets that skip the LDS-resident
    phases (the continuation masks of the reference-cell cache, single stages)."""
    from infercnv_amd import synth
    G, C = 1200, 1500                                     # 256 CUs -> about six cells per workgroup
    x, cs = synth.make_matrix_np(G, C)
    x = x - 1.0
    refs = [np.arange(0, 40, dtype=np.int32), np.arange(40, 90, dtype=np.int32)]
    got = to_host(dev.smooth_chain(to_dev(x), cs, refs, stage_mask=mask)[0])
    v = x.copy()
    if mask & 0x01: v = oc.subtract_ref_expr_from_obs(v, refs)
    if mask & 0x02: v = oc.apply_max_threshold_bounds(v, 3.0)
    if mask & 0x04: v = oc.smooth_by_chromosome(v, cs, 101)
    if mask & 0x08: v = oc.center_columns(v, "median")
    if mask & 0x10: v = oc.subtract_ref_expr_from_obs(v, refs)
    if mask & 0x20: v = oc.invert_log2(v)
    if mask & 0x40:
        mu, sdv = oc.denoise_params(v, np.concatenate(refs), 1.5)
        check_denoise_flips(got, oc.denoise_apply(v, mu, sdv), v, mu, sdv, tol=1e-11, label=f"many cells {mask:#x}")
    else:
        assert np.abs(got - v).max() < 1e-11 * max(1.0, np.abs(v).max())


def _random_chr_layout(G, n_chr, rng, with_single=True):
    cuts = np.sort(rng.choice(np.arange(1, G), size=n_chr - 1, replace=False))
    cs = np.concatenate([[0], cuts, [G]]).astype(np.int32)
    if with_single:
        cs[2] = cs[1] + 1                                   # a single-gene chromosome (left untouched by step 10)
    return cs


@pytest.mark.parametrize("taps", [False, True])
@pytest.mark.parametrize("G,C,n_chr,kw", [(24000, 20, 30, {}), (19001, 9, 24, {"window_length": 151}),
                                          (30000, 6, 2, {"stage_mask": 0x3F}), (26000, 8, 40, {"stage_mask": 0x8F}),
                                          (21000, 7, 25, {"use_bounds": False, "max_thresh": None}),
                                          (18000, 300, 22, {}), (20000, 300, 22, {}), (33001, 5, 60, {}), (40000, 4, 50, {"window_length": 31})])
def test_chain_large_gene_sets_vs_oracle(dev, G, C, n_chr, kw, taps, monkeypatch):
    """Gene sets beyond the fused kernel's LDS-resident limit.  Round 6: the TWO-PASS chain -- pass 1 = steps 8 - 10 by the fused
    kernel on groups of whole chromosomes (strided views of the matrix, chain_w11s.hip), pass 2 = the centre and steps 12 - 22 from
    registers (chain_large.hip: large_center_finish_kernel) -- and, with ICNV_CHAIN_LARGE_TAPS=1 or when a chromosome alone exceeds a
    view (the 14 000 / 16 000-gene case) or the row exceeds 32 768 genes, the three-pass chain of rounds 1 - 5.  18 000 and 20 000
    genes in the bench's chromosome layout: the sizes profiles/r05_sweep.json showed at 14 x the 10 000-gene per-cell cost."""
    from infercnv_amd import synth
    if taps:
        monkeypatch.setenv("ICNV_CHAIN_LARGE_TAPS", "1")
    rng = np.random.default_rng(G)
    x = rng.normal(0.0, 1.0, size=(G, C)) + rng.normal(0.0, 0.5, size=(G, 1))
    if n_chr == 22:
        cs = synth.make_matrix_np(G, 1)[1]                  # the bench's chromosome layout
    else:
        cs = _random_chr_layout(G, n_chr, rng) if n_chr > 2 else np.array([0, 14000, G], dtype=np.int32)
    refs = [np.array([1, 0], dtype=np.int32), np.arange(2, max(3, C // 3), dtype=np.int32)]
    out, pre = dev.smooth_chain(to_dev(x), cs, refs, want_pre_denoise=True, **kw)
    okw = {k: v for k, v in kw.items() if k != "stage_mask"}
    if kw.get("stage_mask") == 0x8F:
        v = oc.subtract_ref_expr_from_obs(x, refs)
        v = oc.apply_max_threshold_bounds(v, 3.0)
        v = oc.smooth_by_chromosome(v, cs, 101)
        want = oc.center_columns(v, "mean")
        assert np.abs(to_host(out) - want).max() < 1e-11
        return
    want, want_pre, musd = oc.smooth_chain(x, cs, refs, want_pre_denoise=True, **kw)
    assert np.abs(to_host(pre) - want_pre).max() < 1e-11 * max(1.0, np.abs(want_pre).max())
    if kw.get("stage_mask", 0x7F) & 0x40:
        check_denoise_flips(to_host(out), want, want_pre, *musd, tol=1e-11, label=f"three-pass G={G}")
    else:
        assert np.abs(to_host(out) - want).max() < 1e-11 * max(1.0, np.abs(want).max())


def test_chain_three_pass_equals_fused(dev, monkeypatch):
    """The three-pass chain forced on a size the fused kernel serves: two independent GPU implementations of the
    same steps must agree (to rounding: different summation orders in the pyramid)."""
    from infercnv_amd import synth
    G, C = 10000, 300
    x, cs = synth.make_matrix_np(G, C)
    refs, _ = synth.groups(C)
    xd = to_dev(x)
    out_f, pre_f = dev.smooth_chain(xd, cs, refs, want_pre_denoise=True)
    monkeypatch.setenv("ICNV_CHAIN_LARGE", "1")
    monkeypatch.setenv("ICNV_CHAIN_LARGE_TAPS", "1")        # the (2T + 1)-tap pass: an implementation that shares no smoothing code with the fused kernel
    out_l, pre_l = dev.smooth_chain(xd, cs, refs, want_pre_denoise=True)
    monkeypatch.delenv("ICNV_CHAIN_LARGE_TAPS")
    out_2, pre_2 = dev.smooth_chain(xd, cs, refs, want_pre_denoise=True)   # the two-pass form (round 6) on the same size
    monkeypatch.delenv("ICNV_CHAIN_LARGE")
    assert (pre_f - pre_l).abs().max().item() < 1e-12
    assert (pre_2 - pre_l).abs().max().item() < 1e-12 and (pre_2 - pre_f).abs().max().item() < 1e-12
    _, ref_pre, (mu, s) = oc.smooth_chain(x, cs, refs, want_pre_denoise=True)
    # two GPU implementations against each other: the fused kernel's output as `got`, the three-pass one as the reference
    check_denoise_flips(to_host(out_f), to_host(out_l), to_host(pre_l), mu, s, tol=1e-12, label="fused vs three-pass")


def test_chain_options(dev):
    """use_bounds=FALSE, no threshold, short / long / no window, fixed-threshold denoise, no denoise."""
    from infercnv_amd import synth
    G, C = 2500, 40
    x, cs = synth.make_matrix_np(G, C)
    refs = [np.arange(0, 6, dtype=np.int32)]
    xd = to_dev(x)
    for kw in ({"use_bounds": False}, {"max_thresh": None}, {"window_length": 3}, {"window_length": 301},
               {"window_length": 1}, {"noise_filter": 0.1}, {"noise_filter": 0.0}, {"stage_mask": 0x3F}):
        out, _ = dev.smooth_chain(xd, cs, refs, **kw)
        okw = dict(kw)
        if "noise_filter" in okw and okw["noise_filter"] is not None:
            okw["noise_filter"] = float(okw["noise_filter"])
        want, want_pre, musd = oc.smooth_chain(x, cs, refs, want_pre_denoise=True, **okw)
        got = to_host(out)
        if kw.get("stage_mask", 0x7F) & 0x40 and kw.get("noise_filter", None) != 0.0:
            check_denoise_flips(got, want, want_pre, *musd, tol=1e-11, label=str(kw))
        else:
            assert np.abs(got - want).max() < 1e-11, kw


def test_chain_median_edge_cases(dev):
    """Median select: ties, all-equal, many exact zeros, even/odd G, heavy outliers."""
    rng = np.random.default_rng(7)
    for G in (2, 3, 64, 1001, 4096):
        cols = [np.zeros(G), np.ones(G) * 3.5, rng.normal(size=G), np.round(rng.normal(size=G), 1),
                np.where(rng.random(G) < 0.7, 0.0, rng.normal(size=G)), rng.normal(size=G) * 1e-300,
                np.concatenate([rng.normal(size=G - 1), [1e300]]), np.arange(G, dtype=float)]
        x = np.stack(cols, axis=1)
        cs = np.array([0, G], dtype=np.int32)
        out, _ = dev.smooth_chain(to_dev(x), cs, [np.array([0], dtype=np.int32)], stage_mask=0x08)
        want = onp.center_columns(x, "median")
        np.testing.assert_array_equal(to_host(out), want)


def test_chain_median_borrowed_range(dev):
    """The median select bins a cell with the value range of the last cell its workgroup measured and measures again
    only when the middle rank falls into the lowest bin.  Many cells per workgroup whose ranges have nothing to do with
    each other: shifted far below / above the previous cell, six orders of magnitude narrower or wider, constant cells,
    ties at the median, values piled up at one end -- the medians must be exact whatever range was borrowed."""
    rng = np.random.default_rng(11)
    for G in (1001, 4096):
        C = 4096
        shift = rng.choice([-1e6, -50.0, -1.0, 0.0, 0.0, 1.0, 50.0, 1e6], size=C) * rng.random(C)
        scale = 10.0 ** rng.integers(-6, 4, size=C)
        x = rng.normal(size=(G, C)) * scale + shift
        kinds = rng.integers(0, 8, size=C)
        x[:, kinds == 0] = np.round(x[:, kinds == 0], 0)                          # heavy ties
        x[:, kinds == 1] = shift[kinds == 1]                                      # constant cells
        m2 = kinds == 2                                                           # most values piled at the lower end
        x[:, m2] = np.where(rng.random((G, int(m2.sum()))) < 0.8, x[:, m2].min(axis=0), x[:, m2])
        m3 = kinds == 3                                                           # ... at the upper end
        x[:, m3] = np.where(rng.random((G, int(m3.sum()))) < 0.8, x[:, m3].max(axis=0), x[:, m3])
        cs = np.array([0, G // 3, G], dtype=np.int32)
        out, _ = dev.smooth_chain(to_dev(x), cs, [np.array([0], dtype=np.int32)], stage_mask=0x08)
        np.testing.assert_array_equal(to_host(out), onp.center_columns(x, "median"))


def test_chain_median_paths_full_chain(dev):
    """The median select inside the full-chain kernels on cells built to take its rare paths after smoothing: thousands
    of exact zeros around the median (far more than the 1 024 candidates ranked directly: the refinement levels and
    the all-members-equal branch), a constant cell, a step profile whose two middle values lie in different bins, plateaus,
    NaN / Inf cells (they must terminate and must not disturb their neighbours)."""
    from infercnv_amd import synth
    G, C = 6000, 64
    cs = synth.chr_layout(G)
    rng = np.random.default_rng(21)
    x = rng.normal(0.0, 0.3, size=(G, C))
    x[:, :4] = 0.0                                            # reference cells: all zero -> steps 8 / 12 subtract 0
    x[:, 4] = 0.0                                             # constant cell
    x[:, 5] = np.concatenate([-np.ones(2000), np.zeros(2500), np.ones(1500)])          # > 1 024 exact zeros at the median
    x[:, 6] = np.concatenate([-np.ones(1000), np.zeros(2000), rng.normal(2.0, 0.1, size=3000)])   # lower middle = last zero
    x[:, 7] = np.where(rng.random(G) < 0.6, 0.0, rng.normal(size=G))                   # zeros mixed with noise
    x[:, 8] = np.repeat(rng.normal(size=G // 200), 200)                                # plateaus: near-ties after smoothing
    clean = x.copy()
    x[17, 9] = np.nan
    x[4000, 10] = np.inf
    x[100, 11] = -np.inf
    x[5999, 11] = np.nan
    refs = [np.arange(4, dtype=np.int32)]
    for mask in (0x7F, 0x3F):
        out, pre = dev.smooth_chain(to_dev(x), cs, refs, stage_mask=mask, want_pre_denoise=True)
        want_out, want_pre, musd = oc.smooth_chain(clean, cs, refs, want_pre_denoise=True, stage_mask=mask)
        if mask == 0x3F:
            want_pre = want_out                               # no denoise stage: the chain's output is the HMM input
        got = to_host(pre)
        ok = np.ones(C, dtype=bool)
        ok[9:12] = False                                      # the non-finite cells: only required to terminate
        assert np.abs(got[:, ok] - want_pre[:, ok]).max() < 1e-12
        assert np.isfinite(got[:, ok]).all()
        if mask == 0x7F:
            check_denoise_flips(to_host(out)[:, ok], want_out[:, ok], want_pre[:, ok], *musd, tol=1e-11,
                                label="median paths, full chain")
    # a few thousand ordinary cells, even G: the "upper middle beyond the ranked bin" branch occurs in ~3 % of them
    G2, C2 = 10000, 3000
    x2, cs2 = synth.make_matrix_np(G2, C2)
    refs2, _ = synth.groups(C2)
    _, pre2 = dev.smooth_chain(to_dev(x2), cs2, refs2, stage_mask=0x3F, want_pre_denoise=True)
    _, want2, _ = oc.smooth_chain(x2, cs2, refs2, want_pre_denoise=True, stage_mask=0x3F)
    assert np.abs(to_host(pre2) - want2).max() < 1e-11


def test_subtract_ref_inv_log(dev):
    """subtract_ref_expr_from_obs(inv_log=TRUE): group means as log2(mean(2^x - 1) + 1) (R/inferCNV_ops.R:1714-1717),
    with and without bounds, several reference groups; only as the stand-alone step."""
    from infercnv_amd import _lib, synth
    G, C = 1500, 90
    x, cs = synth.make_matrix_np(G, C)
    refs = [np.arange(0, 9, dtype=np.int32), np.array([40, 12, 77, 30], dtype=np.int32), np.array([55], dtype=np.int32)]
    for use_bounds in (True, False):
        got, _ = dev.smooth_chain(to_dev(x), cs, refs, stage_mask=_lib.ST_SUBTRACT_REF_1, use_bounds=use_bounds, inv_log=True)
        want = onp.subtract_ref_expr_from_obs(x, refs, inv_log=True, use_bounds=use_bounds)
        assert np.abs(to_host(got) - want).max() < 1e-12
        wantc = oc.subtract_ref_expr_from_obs(x, refs, inv_log=True, use_bounds=use_bounds)
        assert np.abs(to_host(got) - wantc).max() < 1e-12
    plain, _ = dev.smooth_chain(to_dev(x), cs, refs, stage_mask=_lib.ST_SUBTRACT_REF_1)
    assert np.abs(to_host(plain) - to_host(got)).max() > 1e-3            # it is a different statistic
    with pytest.raises(_lib.IcnvError):                                   # run() never combines it with other stages
        dev.smooth_chain(to_dev(x), cs, refs, stage_mask=_lib.ST_ALL, inv_log=True)


def test_chain_many_reference_groups(dev):
    """The first reference round sums every group in one launch (workgroup = gene tile x one of S cell splits of a group,
    S = 256 / n_groups capped at 32): many groups of very different sizes, single-cell groups, odd and even gene counts."""
    from infercnv_amd import synth
    rng = np.random.default_rng(5)
    for G, C, ng in ((2000, 700, 37), (1501, 400, 9), (3000, 900, 130)):
        x, cs = synth.make_matrix_np(G, C)
        perm = rng.permutation(C)[: C // 2]
        cuts = np.sort(rng.choice(np.arange(1, perm.size), size=ng - 1, replace=False))
        refs = [g.astype(np.int32) for g in np.split(perm, cuts)]
        assert min(len(r) for r in refs) >= 1 and len(refs) == ng
        for kw in ({"stage_mask": 0x01}, {"stage_mask": 0x01, "use_bounds": False}):
            got = to_host(dev.smooth_chain(to_dev(x), cs, refs, **kw)[0])
            want = oc.subtract_ref_expr_from_obs(x, refs, use_bounds=kw.get("use_bounds", True))
            assert np.abs(got - want).max() < 1e-12, (G, ng, kw)
    out, pre = dev.smooth_chain(to_dev(x), cs, refs, want_pre_denoise=True)          # the whole chain behind it
    _, want_pre, _ = oc.smooth_chain(x, cs, refs, want_pre_denoise=True)
    assert np.abs(to_host(pre) - want_pre).max() < 1e-11


def test_chain_no_reference_cells_uses_all_observations(dev):
    """R/inferCNV_ops.R:1686-1688: without references all cells form one proxy group (host mirror)."""
    from infercnv_amd import GeneOrder, InfercnvObject, ops, synth
    G, C = 1200, 24
    x, cs = synth.make_matrix_np(G, C)
    chr_names = np.repeat(np.array([f"chr{i + 1}" for i in range(len(cs) - 1)]), np.diff(cs))
    obj = InfercnvObject(expr_data=x, gene_order=GeneOrder(chr=chr_names),
                         observation_grouped_cell_indices={"a": np.arange(0, 10), "b": np.arange(10, C)})
    got = ops.subtract_ref_expr_from_obs(obj).expr_data
    want = oc.subtract_ref_expr_from_obs(x, [np.arange(C, dtype=np.int32)])
    assert np.abs(got - want).max() < 1e-12


# ------------------------------------------------------------------ HMM
def _hmm_input(G, C, seed=0):
    from infercnv_amd import synth
    x, cs = synth.make_matrix_np(G, C, seed=synth.SEED + seed)
    refs, _ = synth.groups(C)
    _, pre, _ = oc.smooth_chain(x, cs, refs, want_pre_denoise=True)
    return pre, cs


@pytest.mark.parametrize("G,C", [(10000, 130), (4613, 64), (1003, 65), (257, 7)])
def test_viterbi_cells_i6_bit_exact(dev, G, C):
    from infercnv_amd import synth
    pre, cs = _hmm_input(G, C, seed=G)
    means, sd, logPi, logDelta = synth.hmm_params_i6()
    st, bad = dev.viterbi_cells(to_dev(pre), cs, means, sd, logPi, logDelta)
    want, wbad = oc.viterbi_cells(pre, cs, means, sd, logPi, logDelta)
    assert int(bad.item()) == 0 and wbad == 0
    np.testing.assert_array_equal(to_host(st), want)
    assert len(np.unique(want)) >= 3


def test_viterbi_fast_path_equals_exact_kernel_and_oracle(dev):
    """The certified fast path (table scores + decision-margin test + exact redo of flagged sequences) must
    return the exact kernel's states bit for bit -- on ordinary data (few flags) and on data with foreign
    values (outside the table's domain, NaN: those sequences are flagged and redone)."""
    from infercnv_amd import synth
    pre, cs = _hmm_input(10000, 1061, seed=21)
    means, sd, logPi, logDelta = synth.hmm_params_i6()
    rng = np.random.default_rng(3)
    foreign = pre.copy()
    gi, ci = rng.integers(0, pre.shape[0], 300), rng.integers(0, pre.shape[1], 300)
    foreign[gi, ci] = rng.choice([np.nan, 1e6, -5.0, 40.0, np.inf], size=300)
    try:
        for x, min_flagged in ((pre, 0), (foreign, 250)):
            xd = to_dev(x)
            dev.viterbi_set_mode(0)
            st_fast, bad = dev.viterbi_cells(xd, cs, means, sd, logPi, logDelta)
            stats = dev.viterbi_last_stats()
            assert stats["path"] == "fast" and stats["sequences"] == 22 * 1061 and stats["table_intervals"] > 100
            assert min_flagged <= stats["flagged"] <= max(2 * min_flagged, 0.01 * stats["sequences"])
            dev.viterbi_set_mode(1)
            st_exact, bad1 = dev.viterbi_cells(xd, cs, means, sd, logPi, logDelta)
            assert dev.viterbi_last_stats()["path"] == "exact"
            assert torch.equal(st_fast, st_exact) and int(bad.item()) == int(bad1.item())
            if x is pre:
                want, _ = oc.viterbi_cells(pre, cs, means, sd, logPi, logDelta)
                np.testing.assert_array_equal(to_host(st_fast), want)
    finally:
        dev.viterbi_set_mode(0)


def test_viterbi_flag_share_decides_redo_or_exact_fallback_on_device(dev):
    """A column batch with at most 2 % of its sequences flagged goes to the wave-per-sequence redo kernel, a batch with
    more is recomputed by the exact kernel -- decided on the device from the batch's own count, so the same call
    gives the same path whatever ran before it (the round-1 library carried a process-global 'skip the fast path for
    the next 8 calls' counter).  States are the exact kernel's either way."""
    from infercnv_amd import synth
    pre, cs = _hmm_input(2000, 640, seed=33)
    means, sd, logPi, logDelta = synth.hmm_params_i6()
    nseq = 22 * 640
    heavy = pre.copy()
    heavy[7, ::4] = np.nan                      # one flagged sequence in every fourth cell: 160 of 14 080 = 1.1 % ...
    heavy[1990, ::2] = 1e9                      # ... plus one in every second: 2.3 % + 1.1 % > 2 %
    light = pre.copy()
    light[7, ::8] = np.nan                      # 80 of 14 080 = 0.6 %
    want = {}
    dev.viterbi_set_mode(1)
    for name, x in (("heavy", heavy), ("light", light)):
        want[name], _ = dev.viterbi_cells(to_dev(x), cs, means, sd, logPi, logDelta)
        assert dev.viterbi_last_stats()["path"] == "exact"
    dev.viterbi_set_mode(0)
    for name, x, fb in (("heavy", heavy, True), ("light", light, False), ("heavy", heavy, True), ("light", light, False)):
        st, _ = dev.viterbi_cells(to_dev(x), cs, means, sd, logPi, logDelta)
        stats = dev.viterbi_last_stats()
        assert stats["path"] == "fast" and stats["sequences"] == nseq and stats["fallback"] is fb, (name, stats)
        assert (stats["flagged"] > 0.02 * nseq) is fb
        assert torch.equal(st, want[name])
    o, _ = oc.viterbi_cells(heavy[:, :8], cs, means, sd, logPi, logDelta)
    np.testing.assert_array_equal(to_host(want["heavy"])[:, :8], o)


@pytest.mark.parametrize("seed", list(range(10)))
def test_viterbi_fast_path_random_models(dev, seed):
    """Random HMMs (3 or 6 states, random increasing means with gaps of 0.3 .. 4 sd, random sd and transition
    probability) on data built to stress the certified path -- values at the state means and mid-points, at the
    table's domain edges, foreign values -- certified fast path against the exact kernel, bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    K = 6 if seed % 2 == 0 else 3
    sd = float(rng.uniform(0.02, 0.5))
    means = 1.0 + np.cumsum(rng.uniform(0.3, 4.0, size=K) * sd)
    means -= means[K // 2] - 1.0
    t = float(rng.choice([1e-6, 1e-4, 1e-2, 0.08]))
    Pi = np.full((K, K), t)
    np.fill_diagonal(Pi, 1 - 5 * t)                      # the reference's (non-stochastic for K = 3) diagonal
    delta = np.full(K, t)
    delta[K // 2] = 1 - 5 * t
    sizes = rng.integers(1, 400, size=int(rng.integers(3, 9)))
    sizes[0] = 1
    G, C = int(sizes.sum()), 256
    cs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    centre = rng.choice(means, size=(1, C))
    x = centre + rng.normal(0.0, sd * rng.uniform(0.2, 1.5), size=(G, C))
    special = rng.random((G, C))
    mids = (means[:-1] + means[1:]) / 2
    pool = np.concatenate([means, mids, means + 14 * sd, means - 14 * sd, [means[0] - 40 * sd, means[-1] + 40 * sd]])
    x = np.where(special < (0.05 if seed < 5 else 0.003), rng.choice(pool, size=(G, C)), x)   # sparse for half the seeds
    foreign_cols = rng.random(C) < 0.04               # foreign values in a few sequences only (each is redone exactly)
    x = np.where((special > 0.97) & foreign_cols[None, :], rng.choice([np.nan, np.inf, -np.inf, 1e9, -1e9], size=(G, C)), x)
    xd = to_dev(x)
    try:
        dev.viterbi_set_mode(0)
        st_fast, bad0 = dev.viterbi_cells(xd, cs, means, sd, np.log(Pi), np.log(delta))
        stats = dev.viterbi_last_stats()
        dev.viterbi_set_mode(2)                              # the register kernel with the full table alone
        st_reg, bad2 = dev.viterbi_cells(xd, cs, means, sd, np.log(Pi), np.log(delta))
        stats_reg = dev.viterbi_last_stats()
        dev.viterbi_set_mode(1)
        st_exact, bad1 = dev.viterbi_cells(xd, cs, means, sd, np.log(Pi), np.log(delta))
    finally:
        dev.viterbi_set_mode(0)
    assert stats["path"] == "fast" and stats_reg["path"] == "fast" and stats_reg["kernel"] in ("register", "exact")
    assert stats["flagged"] < 0.85 * stats["sequences"]      # a good share of the answers comes from the certified path itself
    assert torch.equal(st_fast, st_exact) and int(bad0.item()) == int(bad1.item())
    assert torch.equal(st_reg, st_exact) and int(bad2.item()) == int(bad1.item())
    # a sample of columns against the CPU oracle as well
    pick = np.arange(0, C, 37)
    want, _ = oc.viterbi_cells(x[:, pick], cs, means, sd, np.log(Pi), np.log(delta))
    np.testing.assert_array_equal(st_fast.cpu().numpy().T[:, pick], want)


def test_viterbi_staged_kernel_and_its_full_table_retry(dev):
    """Round 5: the staged fast kernel (observations by whole cache lines through LDS-DMA, a 256-record table whose tails
    end 4.9 sd beyond the outer means for the bench's HMM) runs first; a batch in which more than 2 % of the sequences
    leave that table is redone by the register kernel with the full table (tails of 19 sd), fewer go to the redo kernel
    one by one -- decided on the device.  States are the exact kernel's in every case; the column count is not a multiple
    of 64 (the last group of columns is served as the LAST 64 columns, overlapping its predecessor)."""
    from infercnv_amd import synth
    pre, cs = _hmm_input(10000, 1061, seed=5)
    means, sd, logPi, logDelta = synth.hmm_params_i6()
    nseq = 22 * 1061
    far = means[-1] + 8.0 * sd                      # beyond the staged table, well inside the full one
    rng = np.random.default_rng(8)
    few, many = pre.copy(), pre.copy()
    few[rng.integers(0, 10000, 40), rng.choice(1061, 40, replace=False)] = far            # <= 40 sequences: 0.2 %
    cols = rng.choice(1061, 400, replace=False)
    for c in cols: many[rng.integers(0, 10000, 3), c] = far                                # ~1 100 sequences: 5 %
    many[123, cols[:5]] = np.nan                                                           # ... five of them beyond any table
    want_pre, _ = oc.viterbi_cells(pre, cs, means, sd, logPi, logDelta)
    for name, x, kernel in (("plain", pre, "staged"), ("few", few, "staged"), ("many", many, "staged+register")):
        xd = to_dev(x)
        dev.viterbi_set_mode(0)
        st0, _ = dev.viterbi_cells(xd, cs, means, sd, logPi, logDelta)
        s0 = dev.viterbi_last_stats()
        dev.viterbi_set_mode(2)
        st2, _ = dev.viterbi_cells(xd, cs, means, sd, logPi, logDelta)
        s2 = dev.viterbi_last_stats()
        dev.viterbi_set_mode(1)
        st1, _ = dev.viterbi_cells(xd, cs, means, sd, logPi, logDelta)
        assert s0["path"] == "fast" and s0["kernel"] == kernel and not s0["fallback"], (name, s0)
        assert s2["kernel"] == "register" and s2["table_intervals"] > 600, (name, s2)
        assert s0["table_intervals"] < 300 and s0["sequences"] == nseq
        if name == "few": assert 30 <= s0["flagged"] <= 60 and s2["flagged"] <= 5
        if name == "many": assert 5 <= s0["flagged"] <= 20 and s0["flagged"] == s2["flagged"]     # what the full table leaves: the NaNs (+ chance ties)
        assert torch.equal(st0, st1) and torch.equal(st2, st1), name
        if name == "plain": np.testing.assert_array_equal(to_host(st0), want_pre)
    dev.viterbi_set_mode(0)


def test_viterbi_column_batches(dev, monkeypatch):
    """The back-pointer scratch bounds the columns of one launch (4 GiB by default: 214 000 cells at 10 000 genes);
    more cells run as several column batches.  A 1 MiB budget splits 700 cells x 1 500 genes into batches of 128
    columns (the scratch is sized for the exact kernel's 4-byte words on either path); every batch, the partial last one and the
    foreign values that send sequences of different batches to the redo kernel must give the single-launch states."""
    from infercnv_amd import synth
    pre, cs = _hmm_input(1500, 700, seed=9)
    pre[37, 5] = np.nan                  # a flagged sequence in the first batch ...
    pre[900, 650] = 1e9                  # ... and one in the last
    means, sd, logPi, logDelta = synth.hmm_params_i6()
    xd = to_dev(pre)
    ref, bad_ref = dev.viterbi_cells(xd, cs, means, sd, logPi, logDelta)
    assert dev.viterbi_last_stats()["path"] == "fast"
    want, wbad = oc.viterbi_cells(pre, cs, means, sd, logPi, logDelta)
    np.testing.assert_array_equal(to_host(ref), want)
    monkeypatch.setenv("ICNV_VITERBI_SCRATCH_MB", "1")
    try:
        for mode in (0, 1):
            dev.viterbi_set_mode(mode)
            st, bad = dev.viterbi_cells(xd, cs, means, sd, logPi, logDelta)
            assert dev.viterbi_last_stats()["path"] == ("fast" if mode == 0 else "exact")
            assert torch.equal(st, ref) and int(bad.item()) == int(bad_ref.item()) == wbad
    finally:
        dev.viterbi_set_mode(0)


def test_viterbi_adversarial_near_ties_bit_exact(dev):
    """Inputs sitting on emission-branch boundaries and state mid-points, 1-gene and 2-gene chromosomes."""
    from infercnv_amd import synth
    means, sd, logPi, logDelta = synth.hmm_params_i6()
    rng = np.random.default_rng(5)
    sizes = [1, 2, 3, 700, 1, 64, 129, 100]
    G, C = sum(sizes), 192
    cs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    mids = (means[:-1] + means[1:]) / 2
    pool = np.concatenate([means, mids, means + 0.67448975 * sd, means - 0.67448975 * sd,
                           means + 5.656854249492380 * sd, [0.0, 1.0, 10.0, -3.0, 1e-300]])
    x = rng.choice(pool, size=(G, C)) + rng.choice([0.0, 1e-16, -1e-16, 1e-12, 1e-9], size=(G, C))
    x[:, :32] = rng.normal(1.0, 0.15, size=(G, 32))
    for t in (1e-6, 1e-2, 0.1):
        Pi, delta = onp.get_HMM_i6(t)
        st, _ = dev.viterbi_cells(to_dev(x), cs, means, sd, np.log(Pi), np.log(delta))
        want, _ = oc.viterbi_cells(x, cs, means, sd, np.log(Pi), np.log(delta))
        np.testing.assert_array_equal(to_host(st), want)
    # asymmetric (non-uniform) transition matrix exercises every back-pointer value
    Pi = rng.dirichlet(np.ones(6), size=6)
    delta = rng.dirichlet(np.ones(6))
    st, _ = dev.viterbi_cells(to_dev(x), cs, means, sd, np.log(Pi), np.log(delta))
    want, _ = oc.viterbi_cells(x, cs, means, sd, np.log(Pi), np.log(delta))
    np.testing.assert_array_equal(to_host(st), want)


def test_viterbi_extreme_operands_bit_exact(dev):
    """Observations / parameters outside the range where the kernel's lean division is proven identical to
    IEEE division (|x| >= 2^40, subnormals, a zero state mean, a huge sd) take the plain IEEE path."""
    from infercnv_amd import synth
    means, sd, logPi, logDelta = synth.hmm_params_i6()
    rng = np.random.default_rng(17)
    G, C = 600, 128
    cs = np.array([0, 250, 600], dtype=np.int32)
    x = rng.normal(1.0, 0.3, size=(G, C))
    odd = rng.random((G, C)) < 0.03
    x[odd] = rng.choice([2.0 ** 40, -2.0 ** 40, 1e13, 1e150, -1e200, 1e-310, -4e-320, 0.0, 1e300], size=int(odd.sum()))
    t = 1e-2                                   # a soft transition matrix keeps the scores finite longer
    Pi, delta = onp.get_HMM_i6(t)
    for mm, s_ in ((means, sd), (np.array([0.0, 0.5, 1.0, 1.5, 2.0, 3.0]), sd), (means, 2.0 ** 45), (means, 2.0 ** -41)):
        st, bad = dev.viterbi_cells(to_dev(x), cs, mm, s_, np.log(Pi), np.log(delta))
        want, wbad = oc.viterbi_cells(x, cs, mm, s_, np.log(Pi), np.log(delta))
        np.testing.assert_array_equal(to_host(st), want)
        assert int(bad.item()) == wbad


def test_viterbi_i3_bit_exact(dev):
    pre, cs = _hmm_input(3000, 100, seed=3)
    refs = np.arange(10, dtype=np.int32)
    mu, sigma = oc.mean_sd_of_cells(pre, refs)
    gmu, gsigma = dev.cells_mean_sd(to_dev(pre), refs)
    assert abs(gmu - mu) < 1e-13 and abs(gsigma - sigma) < 1e-13
    delta_m = abs(-1.6448536269514722 * sigma)
    m3 = np.array([mu - delta_m, mu, mu + delta_m])
    Pi, delta = onp.get_HMM_i3(1e-6)
    st, _ = dev.viterbi_cells(to_dev(pre), cs, m3, sigma, np.log(Pi), np.log(delta))
    want, _ = oc.viterbi_cells(pre, cs, m3, sigma, np.log(Pi), np.log(delta))
    np.testing.assert_array_equal(to_host(st), want)


def test_viterbi_groups_and_broadcast(dev):
    from infercnv_amd import synth
    pre, cs = _hmm_input(4000, 150, seed=9)
    rng = np.random.default_rng(1)
    perm = rng.permutation(150)
    groups = [perm[:40], perm[40:41], perm[41:120]]          # cells 120.. are in no group
    means, _, logPi, logDelta = synth.hmm_params_i6()
    sds = [0.08, 0.17, 0.05]
    xd = to_dev(pre)
    st, bad = dev.viterbi_groups(xd, cs, groups, means, sds, logPi, logDelta)
    gm = dev.group_means(xd, groups).cpu().numpy().T          # (G, n_groups)
    # the library returns the CORRECTLY ROUNDED mean (double-double accumulation): pinned against exact rational
    # arithmetic, whatever the order of the cells ...
    from fractions import Fraction
    for q, g in enumerate(groups):
        for gene in (0, 1, 77, 1999, 3999):
            exact = sum((Fraction(float(v)) for v in pre[gene, g]), Fraction(0)) / len(g)
            assert gm[gene, q] == float(exact), (q, gene)
    # ... and R's rowMeans (LDOUBLE accumulation: x87 on this host, as restated by both oracles) differs from it by at
    # most one unit in the last place, in a small share of the genes
    r_gm = oc.group_means(pre, groups)
    np.testing.assert_array_equal(r_gm, onp.group_means(pre, groups))
    ulp = np.spacing(np.abs(r_gm))
    assert (np.abs(gm - r_gm) <= ulp).all()
    print(f"[group means] {(gm != r_gm).sum()} of {gm.size} means differ from the x87 rowMeans in the last bit")
    got = to_host(st)
    for q, g in enumerate(groups):
        # identical inputs (the GPU's own group means) -> bit-exact trace, broadcast to every member
        want, _ = oc.viterbi_cells(gm[:, q:q + 1], cs, means, sds[q], logPi, logDelta)
        for c in g:
            np.testing.assert_array_equal(got[:, c], want[:, 0])
    assert (got[:, perm[120:]] == 255).all()
    # end to end against the oracle (R's own rowMeans arithmetic): bit-exact states
    full, _ = oc.viterbi_groups(pre, cs, groups, means, sds, logPi, logDelta)
    member = np.concatenate(groups)
    np.testing.assert_array_equal(got[:, member], full[:, member])
    proxy = dev.states_to_proxy(st, 6)
    np.testing.assert_array_equal(to_host(proxy)[:, member], oc.states_to_proxy(full, 6)[:, member])


def test_viterbi_long_chromosomes(dev):
    """The wave-per-sequence Viterbi (group modes, few columns, flagged sequences) keeps a sequence's back-pointers in
    LDS up to 4 096 genes per chromosome and in a global scratch beyond: chromosomes of 4 096 and 4 097 genes next to a
    short one, per-cell (< 64 columns: wave-per-sequence kernel), per-cell on the certified fast path with flagged
    sequences (redo kernel), and group mode -- all against the oracle, bit for bit."""
    from infercnv_amd import synth
    sizes = [4096, 60, 4097]
    G = sum(sizes)
    cs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    rng = np.random.default_rng(17)
    means, sd, logPi, logDelta = synth.hmm_params_i6()
    for C in (5, 96):
        state = rng.choice(means, size=(1, C)) + np.where(rng.random((G, C)) < 0.002, 0.5, 0.0).cumsum(axis=0) % 1.0
        x = state + rng.normal(0.0, 0.12, size=(G, C))
        if C == 96:
            x[100, 3] = np.nan            # flagged on the fast path: redone by the wave-per-sequence kernel
            x[5000, 70] = 1e9
        st, bad = dev.viterbi_cells(to_dev(x), cs, means, sd, logPi, logDelta)
        assert dev.viterbi_last_stats()["path"] == ("fast" if C == 96 else "exact")
        want, wbad = oc.viterbi_cells(x, cs, means, sd, logPi, logDelta)
        np.testing.assert_array_equal(to_host(st), want)
        assert int(bad.item()) == wbad
    groups = [np.arange(0, 30, dtype=np.int32), np.arange(30, 96, dtype=np.int32)]
    xd = to_dev(x)
    stg, _ = dev.viterbi_groups(xd, cs, groups, means, [0.05, 0.04], logPi, logDelta)
    gm = dev.group_means(xd, groups).cpu().numpy().T
    got = to_host(stg)
    for q, g in enumerate(groups):
        want, _ = oc.viterbi_cells(gm[:, q:q + 1], cs, means, [0.05, 0.04][q], logPi, logDelta)
        np.testing.assert_array_equal(got[:, g[0]], want[:, 0])
        np.testing.assert_array_equal(got[:, g[-1]], want[:, 0])


def test_underflow_is_reported(dev):
    """-Inf in the last nu row -> the reference stop()s (R/inferCNV_HMM.R:1165); we return ICNV_ERR_UNDERFLOW."""
    from infercnv_amd import IcnvError, hmm
    x = np.ones(20)
    Pi = np.full((6, 6), 1e-6)
    np.fill_diagonal(Pi, 1 - 5e-6)
    delta = np.zeros(6)                      # log(0) = -Inf everywhere
    with pytest.raises(IcnvError) as e:
        hmm.Viterbi_dthmm_adj(x, Pi, delta, np.linspace(0.4, 1.4, 6), np.full(6, 0.2))
    assert e.value.code == 4


# ------------------------------------------------------------------ median filter
def test_median_filter_exact(dev):
    rng = np.random.default_rng(4)
    sizes = [40, 1, 7, 95, 9]
    G, C = sum(sizes), 61
    cs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    x = rng.normal(size=(G, C))
    x[5:9, :] = 0.0
    perm = rng.permutation(C)
    tiles = [perm[:23], perm[23:24], perm[24:33], perm[33:55]]   # cells 55.. in no tile: copied through
    for w in (3, 7, 9):
        out = dev.median_filter(to_dev(x), cs, tiles, w)
        want = oc.median_filter(x, cs, tiles, w)
        np.testing.assert_array_equal(to_host(out), want)


def test_median_filter_block_boundaries(dev):
    """The 9 x 9 filter cuts every (tile, chromosome) block into tiles (56 x 32 for the classification pass, 32 x 16 for the dense
    pass) and treats interior outputs (dense pass or single outputs) and border outputs (single outputs, clamped windows)
    differently: chromosome lengths and tile sizes around every boundary of those splits (8 | 9, one interior row, odd / even
    numbers of interior cells, one full tile +- 1, two gene blocks +- 1), ties included."""
    rng = np.random.default_rng(14)
    sizes = [8, 9, 10, 17, 36, 40, 41, 73, 3]
    G = sum(sizes)
    cs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    tile_sizes = [8, 9, 10, 11, 24, 25, 40, 41, 2]
    C = sum(tile_sizes) + 3
    x = np.round(rng.normal(size=(G, C)), 1)           # one decimal: many ties
    x[:, ::7] = rng.normal(size=(G, len(range(0, C, 7))))
    perm = rng.permutation(C)
    off = np.concatenate([[0], np.cumsum(tile_sizes)])
    tiles = [perm[off[i]:off[i + 1]] for i in range(len(tile_sizes))]
    out = dev.median_filter(to_dev(x), cs, tiles, 7)
    want = oc.median_filter(x, cs, tiles, 7)
    np.testing.assert_array_equal(to_host(out), want)
    # tiles that cover every cell: the pass-through copy is skipped, every element must still be written
    tiles_all = tiles + [perm[off[-1]:]]
    out2 = dev.median_filter(to_dev(x), cs, tiles_all, 7)
    np.testing.assert_array_equal(to_host(out2), oc.median_filter(x, cs, tiles_all, 7))


@pytest.mark.parametrize("case", ["share_0.3", "share_0.55", "share_0.9", "two_values", "runs_along_genes", "negative_zero", "nan_candidate"])
def test_median_filter_majority_shortcut_is_exact(dev, case):
    """The majority shortcut of the 9 x 9 kernels (a value on more than half of a window's positions is its median: the
    denoised matrix holds one value mu at most of its entries, R/inferCNV_ops.R:2335) must change nothing: matrices with a
    dominant value at shares below, around and far above one half, two dominant values in two halves of the cells (the
    candidate has to be re-seeded), dominant runs along the genes (whole patches decided, whole patches not), borders and
    even window counts included -- exact against the oracle, many workgroups deep (persistent workgroups carry the candidate
    from patch to patch)."""
    rng = np.random.default_rng(len(case))
    sizes = [150, 9, 61, 330, 8, 40, 75]
    G = sum(sizes)
    cs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    tile_sizes = [96, 41, 9, 8, 130, 57]
    C = sum(tile_sizes)
    x = rng.normal(1.0, 0.2, size=(G, C))
    mu = 1.0124904741170890
    if case.startswith("share_"):
        x[rng.random((G, C)) < float(case.split("_")[1])] = mu
    elif case == "two_values":
        m = rng.random((G, C)) < 0.8
        x[:, : C // 2][m[:, : C // 2]] = mu
        x[:, C // 2:][m[:, C // 2:]] = -0.25
    elif case == "runs_along_genes":
        for c in range(C):                                  # per cell: long stretches of mu, interrupted by continuous stretches
            g = 0
            while g < G:
                n = int(rng.integers(20, 200))
                if rng.random() < 0.7:
                    x[g:g + n, c] = mu
                g += n
    elif case == "negative_zero":
        x[rng.random((G, C)) < 0.45] = 0.0
        x[rng.random((G, C)) < 0.45] = -0.0                 # +0.0 and -0.0 are ONE value for the comparison and for the median
    else:
        x[rng.random((G, C)) < 0.8] = mu
        x[75, :] = np.nan                                   # a probe position may hold a NaN: it never equals anything
    perm = rng.permutation(C)
    off = np.concatenate([[0], np.cumsum(tile_sizes)])
    tiles = [perm[off[i]:off[i + 1]] for i in range(len(tile_sizes))]
    if case == "two_values":                                # tiles that stay inside one half, so that whole workgroups see one value
        left, right = np.arange(C // 2), np.arange(C // 2, C)
        tiles = [left[:100], left[100:], right[:77], right[77:]]
    if case == "nan_candidate":
        x[75, :] = 7.0                                      # (the oracle has no NaN policy for the filter: keep the data finite ...)
        x[75, ::2] = mu
    out = dev.median_filter(to_dev(x), cs, tiles, 7)
    want = oc.median_filter(x, cs, tiles, 7)
    np.testing.assert_array_equal(to_host(out), want)


def test_median_filter_slow_list_and_developer_modes(dev, monkeypatch):
    """The three-kernel 9 x 9 scheme has paths a healthy run hardly takes: a tile whose undecided outputs do not fit the
    workgroup's queue segment goes to the SLOW list (every one of its outputs is then computed singly), and the developer
    modes switch the majority test (1) or the queue for interior outputs (2) off.  All of them must give the oracle's matrix.
    Runs in subprocesses: the switches are read once per process."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, numpy as np, torch
sys.path[:0] = [%r, %r + '/oracle']
import oracle_c as oc
from infercnv_amd import device
torch.cuda.set_device(0); device.init(0)
rng = np.random.default_rng(77)
sizes = [150, 9, 61, 330, 8, 40, 75]; G = sum(sizes)
cs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
tsz = [96, 41, 9, 8, 130, 57]; C = sum(tsz)
x = rng.normal(1.0, 0.2, size=(G, C)); x[rng.random((G, C)) < 0.8] = 1.012490474117089
x[:, 200:] = rng.normal(1.0, 0.2, size=(G, C - 200))          # a region without a dominant value
perm = rng.permutation(C); off = np.concatenate([[0], np.cumsum(tsz)])
tiles = [perm[off[i]:off[i + 1]].astype(np.int32) for i in range(len(tsz))]
got = device.median_filter(torch.from_numpy(np.ascontiguousarray(x.T)).cuda(), cs, tiles, 7).cpu().numpy().T
assert np.array_equal(got, oc.median_filter(x, cs, tiles, 7))
print("MF9_OK")
""" % (root, root)
    for env in ({"ICNV_MF9_QCAP": "40"}, {"ICNV_MF9_QCAP": "1"}, {"ICNV_MF9_MODE": "1"}, {"ICNV_MF9_MODE": "2"}, {"ICNV_MF9_MODE": "3"}):
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env), timeout=300)
        assert res.returncode == 0 and "MF9_OK" in res.stdout, (env, res.stdout[-1500:], res.stderr[-1500:])


def test_median_filter_strip_kernel_paths(dev):
    """Round 6: the dense pass of the 9 x 9 filter is the STRIP kernel (32-bit compound keys: a monotone 24-bit bucket code + an
    8-bit element id; exact because an output is only written when no other element of its window shares the median's code, anything
    else is queued for the single-output kernel, and an overflowing queue hands the tiles to the fp64 dense pass behind a gate).  Which
    path a matrix takes depends on what two small probe launches find in a sample -- a dominant value, up to three repeated values
    (codes of their own), the value range, discrete data (strip kernel off) -- but the RESULT must not: every data shape below, under
    every developer switch, equals the oracle's matrix.  Runs in subprocesses: the switches are read once per process."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, numpy as np, torch
sys.path[:0] = [%r, %r + '/oracle']
import oracle_c as oc
from infercnv_amd import device
torch.cuda.set_device(0); device.init(0)
sizes = [150, 9, 61, 330, 8, 40, 75, 64, 65]; G = sum(sizes)
cs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
tsz = [300, 41, 9, 8, 130, 129, 17]; C = sum(tsz)
mu = 1.012490474117089
def data(case, rng):
    x = rng.normal(1.0, 0.2, size=(G, C))
    if case == "continuous":
        pass
    elif case == "dominant_islands":                 # the denoised shape: one value with islands of continuous values
        x[:] = mu
        for _ in range(25):
            g0, c0 = int(rng.integers(0, G)), int(rng.integers(0, C))
            g1, c1 = min(G, g0 + int(rng.integers(1, 200))), min(C, c0 + int(rng.integers(1, 200)))
            x[g0:g1, c0:c1] = rng.normal(1.3, 0.2, size=(g1 - g0, c1 - c0))
    elif case == "rounded":                          # discrete data: the probe turns the strip kernel off
        x = np.round(x, 1)
    elif case == "three_values":                     # repeated values below a majority: codes of their own
        r = rng.random((G, C))
        x[r < 0.10] = 1.0; x[(r >= 0.10) & (r < 0.18)] = mu; x[(r >= 0.18) & (r < 0.24)] = 0.75
    elif case == "six_values":                       # more repeated values than codes: strip kernel off, or its queue takes the rest
        r = rng.random((G, C))
        for k, v in enumerate((1.0, mu, 0.75, 1.25, 0.5, 1.5)):
            x[(r >= 0.05 * k) & (r < 0.05 * (k + 1))] = v
    elif case == "rare_ties":                        # a repeated value the probe is unlikely to see (0.3 %%): its windows are queued
        x[rng.random((G, C)) < 0.003] = 1.0
        x[100:140, 50:90] = 1.0                      # ... and a block where it is every window's median
    elif case == "outliers":                         # end buckets: infinities and huge values, next to ordinary ones
        r = rng.random((G, C))
        x[r < 0.002] = np.inf; x[(r >= 0.002) & (r < 0.004)] = -np.inf; x[(r >= 0.004) & (r < 0.006)] = 1e300; x[(r >= 0.006) & (r < 0.008)] = -1e300
        x[200:260, 100:180] = np.inf                 # windows whose median is +Inf
    elif case == "tight":                            # a range set by two far values, everything else inside a few buckets: collisions everywhere
        x = 1.0 + 1e-9 * rng.normal(size=(G, C))
        x[0, 0], x[1, 1] = -5e3, 5e3
        x[rng.random((G, C)) < 0.001] = 4e3
    return x
rng = np.random.default_rng(606)
perm = rng.permutation(C); off = np.concatenate([[0], np.cumsum(tsz)])
tiles = [perm[off[i]:off[i + 1]].astype(np.int32) for i in range(len(tsz))]
tiles[0] = np.sort(tiles[0])
for case in ("continuous", "dominant_islands", "rounded", "three_values", "six_values", "rare_ties", "outliers", "tight"):
    x = data(case, rng)
    got = device.median_filter(torch.from_numpy(np.ascontiguousarray(x.T)).cuda(), cs, tiles, 7).cpu().numpy().T
    want = oc.median_filter(x, cs, tiles, 7)
    assert np.array_equal(got, want), (case, int((got != want).sum()))
print("MF9_STRIP_OK")
""" % (root, root)
    for env in ({}, {"ICNV_MF9_STRIP": "0"}, {"ICNV_MF9_FQCAP": "0"}, {"ICNV_MF9_FQCAP": "7"}, {"ICNV_MF9_PROBE": "0"}, {"ICNV_MF9_BORDER": "0"}, {"ICNV_MF9_SWEEP": "0"}, {"ICNV_MF9_SWEEP": "1"}, {"ICNV_MF9_SWEEP": "1", "ICNV_MF9_QCAP": "3", "ICNV_MF9_MODE": "2"}, {"ICNV_MF9_SWEEP": "2", "ICNV_MF9_QCAP": "3"}, {"ICNV_MF9_SWEEP": "2", "ICNV_MF9_MODE": "3"}, {"ICNV_MF9_QCAP": "40", "ICNV_MF9_FQCAP": "100"}):
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
        assert res.returncode == 0 and "MF9_STRIP_OK" in res.stdout, (env, res.stdout[-1500:], res.stderr[-1500:])


# ------------------------------------------------------------------ host mirror == device path == oracle
def test_host_mirror_runs_reference_workflow(dev, example):
    from infercnv_amd import GeneOrder, InfercnvObject, hmm, noise_reduction, ops
    levels = example["chr_levels"][example["chr_codes"]]
    obj = InfercnvObject(expr_data=example["log"], gene_order=GeneOrder(chr=levels),
                         reference_grouped_cell_indices={"normal": example["ref_normal"]},
                         observation_grouped_cell_indices={"tumor": example["obs_tumor"]},
                         tumor_subclusters={"subclusters": {"tumor": {"tumor_s1": example["obs_tumor"]},
                                                            "normal": {"normal_s1": example["ref_normal"]}}})
    # run()'s order, one wrapper per step (R/inferCNV_ops.R:771-1031, 1573)
    o = ops.subtract_ref_expr_from_obs(obj)
    o = ops.apply_max_threshold_bounds(o, 3)
    o = ops.smooth_by_chromosome(o, 101)
    o = ops.center_cell_expr_across_chromosome(o, "median")
    o = ops.subtract_ref_expr_from_obs(o)
    o14 = ops.invert_log2(o)
    o22 = ops.clear_noise_via_ref_mean_sd(o14, 1.5)
    _, ref_pre, (mu, s) = oc.smooth_chain(example["log"], example["chr_start"], [example["ref_normal"]], want_pre_denoise=True)
    assert np.abs(o14.expr_data - ref_pre).max() < 1e-12
    check_denoise_flips(o22.expr_data, example["expr_data"], ref_pre, mu, s, tol=1e-11, expect=0, label="step functions vs @expr.data")
    fused, hmm_in = ops.hip_smooth_chain(obj, return_hmm_input=True)
    assert np.abs(hmm_in.expr_data - o14.expr_data).max() < 1e-12
    check_denoise_flips(fused.expr_data, o22.expr_data, ref_pre, mu, s, tol=1e-12, expect=0, label="fused vs step functions")
    # i6 HMM on whole samples + proxy values + median filter
    cnv = {k: {"mean": m, "sd": 0.2} for k, m in zip(hmm.CNV_LEVELS, [0.41, 0.84, 1.017, 1.12, 1.24, 1.44])}
    h = hmm.predict_CNV_via_HMM_on_whole_tumor_samples(hmm_in, True, cnv)
    want, _ = oc.viterbi_groups(hmm_in.expr_data, example["chr_start"], [example["obs_tumor"], example["ref_normal"]],
                                [cnv[k]["mean"] for k in hmm.CNV_LEVELS], [0.2, 0.2], np.log(onp.get_HMM_i6()[0]),
                                np.log(onp.get_HMM_i6()[1]))
    np.testing.assert_array_equal(h.expr_data, want)
    p = hmm.assign_HMM_states_to_proxy_expr_vals(h)
    assert set(np.unique(p.expr_data)) <= {0.0, 0.5, 1.0, 1.5, 2.0, 3.0}
    hc = hmm.predict_CNV_via_HMM_on_indiv_cells(hmm_in, cnv)
    wc, _ = oc.viterbi_cells(hmm_in.expr_data, example["chr_start"], [cnv[k]["mean"] for k in hmm.CNV_LEVELS], 0.2,
                             np.log(onp.get_HMM_i6()[0]), np.log(onp.get_HMM_i6()[1]))
    np.testing.assert_array_equal(hc.expr_data, wc)
    mf = noise_reduction.apply_median_filtering(fused)
    wmf = oc.median_filter(fused.expr_data, example["chr_start"], [example["obs_tumor"], example["ref_normal"]], 7)
    np.testing.assert_array_equal(mf.expr_data, wmf)


def test_noise_logistic_denoise_vs_oracle(dev, example):
    """Step 22 with noise_logistic = TRUE (R/inferCNV_ops.R:2249-2252, 2326-2330 -> depress_log_signal_midpt_val ->
    .apply_logistic_val_adj, R/inferCNV_heatmap.R:2791-2810): the stand-alone step functions on the golden object's step-14
    matrix, the fused chain, and the three-pass chain, against the oracle's element-by-element restatement."""
    from infercnv_amd import GeneOrder, InfercnvObject, ops
    levels = example["chr_levels"][example["chr_codes"]]
    refs = example["ref_normal"]
    obj = InfercnvObject(expr_data=example["log"], gene_order=GeneOrder(chr=levels),
                         reference_grouped_cell_indices={"normal": refs}, observation_grouped_cell_indices={"tumor": example["obs_tumor"]})
    _, pre, (mu, s) = oc.smooth_chain(example["log"], example["chr_start"], [refs], want_pre_denoise=True)
    o14 = obj.copy()
    o14.expr_data = pre
    want = onp.clear_noise_via_ref_mean_sd(pre, refs, 1.5, noise_logistic=True)
    got = ops.clear_noise_via_ref_mean_sd(o14, 1.5, noise_logistic=True).expr_data
    assert np.abs(got - want).max() < 1e-12
    assert np.abs(want - pre).max() > 0.01 and np.abs(want - onp.clear_noise_via_ref_mean_sd(pre, refs, 1.5)).max() > 0.01   # it does something, and not the select
    # every value moves towards the centre and stays on its side of it; the further out, the less it moves
    assert (np.abs(got - mu) <= np.abs(pre - mu) + 1e-15).all()
    assert ((got > mu) == (pre > mu)).all() and ((got < mu) == (pre < mu)).all()
    far = np.abs(pre - mu) > s + 0.3
    assert far.any() and (np.abs(got[far] - pre[far]) < 3e-3 * np.abs(pre[far] - mu)).all()      # 1 - p < exp(-20 * 0.3)
    want_t = onp.clear_noise(pre, refs, 0.1, noise_logistic=True)
    assert np.abs(ops.clear_noise(o14, 0.1, noise_logistic=True).expr_data - want_t).max() < 1e-12
    # fused: steps 8..14 + logistic step 22 in one call; the HMM input is the matrix before step 22
    fused, hmm_in = ops.hip_smooth_chain(obj, return_hmm_input=True, noise_logistic=True)
    assert np.abs(hmm_in.expr_data - pre).max() < 1e-12
    assert np.abs(fused.expr_data - onp.apply_logistic_val_adj(hmm_in.expr_data, mu, s)).max() < 1e-11
    os.environ["ICNV_CHAIN_LARGE"] = "1"
    try:
        f2, h2 = ops.hip_smooth_chain(obj, return_hmm_input=True, noise_logistic=True)
    finally:
        del os.environ["ICNV_CHAIN_LARGE"]
    assert np.abs(f2.expr_data - fused.expr_data).max() < 1e-11 and np.abs(h2.expr_data - pre).max() < 1e-11


def test_split_phase_equals_one_call(dev):
    """The multi-GPU split-phase API on one rank must equal the one-call form, and
    two 'ranks' emulated on one GPU (partials summed by hand) must equal it too."""
    from infercnv_amd import sharded, synth
    G, C = 6000, 96
    x, cs = synth.make_matrix_np(G, C)
    refs, _ = synth.groups(C)
    xd = to_dev(x)
    one, one_pre = dev.smooth_chain(xd, cs, refs, want_pre_denoise=True)
    plan = dev.ChainPlan(G, C, cs, refs)
    out, pre = sharded.ShardedChain(plan).run(xd, want_pre_denoise=True)
    assert torch.equal(out, one) and torch.equal(pre, one_pre)
    # two shards on one device
    bounds = [sharded.shard_bounds(C, 2, r) for r in range(2)]
    plans = [dev.ChainPlan(G, b[1] - b[0], cs, sharded.localize_groups(refs, *b)) for b in bounds]
    shards = [xd[b[0]:b[1]].contiguous() for b in bounds]
    for r in range(plans[0].num_rounds):
        bufs = [p.round_partial(r, s) for p, s in zip(plans, shards)]
        tot = bufs[0] + bufs[1]
        for b in bufs:
            b.copy_(tot)
        for p in plans:
            p.round_finish(r)
    outs = [p.apply(s)[0] for p, s in zip(plans, shards)]
    got = torch.cat(outs, dim=0)
    # (the two shards' partial sums are added in another order than one launch adds them: bounds differ by rounding)
    _, ref_pre, (mu, s) = oc.smooth_chain(x, cs, refs, want_pre_denoise=True)
    check_denoise_flips(to_host(got), to_host(one), ref_pre, mu, s, tol=1e-12, label="two shards vs one call")


# ------------------------------------------------------------------ full-size properties
@pytest.mark.parametrize("C", [50000])
def test_full_size_properties(dev, C):
    """BASELINE.json configs[1]: 10 000 genes x 50 000 cells, smooth + i6 HMM on one GPU."""
    from infercnv_amd import synth
    G = 10000
    x, cs = synth.make_matrix_torch(G, C, "cuda")
    refs, _ = synth.groups(C)
    out, pre = dev.smooth_chain(x, cs, refs, want_pre_denoise=True)
    means, sd, logPi, logDelta = synth.hmm_params_i6()
    st, bad = dev.viterbi_cells(pre, cs, means, sd, logPi, logDelta)
    torch.cuda.synchronize()
    assert int(bad.item()) == 0
    assert torch.isfinite(pre).all() and (pre > 0).all()
    smin, smax = int(st.min()), int(st.max())
    assert 1 <= smin and smax <= 6
    # (1) spot parity: random cells re-done by the oracle given the GPU's own reference statistics
    rng = np.random.default_rng(0)
    pick = np.sort(rng.choice(C, 48, replace=False))
    pre_h = pre[torch.as_tensor(pick, device="cuda")].cpu().numpy().T
    want_st, _ = oc.viterbi_cells(pre_h, cs, means, sd, logPi, logDelta)
    np.testing.assert_array_equal(st[torch.as_tensor(pick, device="cuda")].cpu().numpy().T, want_st)
    # (1b) the certified fast path against the exact kernel over the WHOLE matrix: 1.1 M sequences, 5e8 state calls
    stats = dev.viterbi_last_stats()
    assert stats["path"] == "fast" and stats["sequences"] == 22 * C and not stats["fallback"]
    dev.viterbi_set_mode(1)
    st_exact, bad_exact = dev.viterbi_cells(pre, cs, means, sd, logPi, logDelta)
    assert dev.viterbi_last_stats()["path"] == "exact"
    dev.viterbi_set_mode(0)
    assert torch.equal(st, st_exact) and int(bad_exact.item()) == 0
    del st_exact
    # (2) cells are independent given the reference statistics: recomputing a slice that contains all
    #     reference cells reproduces those columns exactly
    n_ref = sum(len(r) for r in refs)
    sub = x[: n_ref + 512].contiguous()
    out2, pre2 = dev.smooth_chain(sub, cs, refs, want_pre_denoise=True)
    assert torch.equal(pre2, pre[: n_ref + 512])
    assert torch.equal(out2, out[: n_ref + 512])
    # (3) centring property: after steps 8-11 every cell's median is exactly removed
    cen, _ = dev.smooth_chain(x[:4096].contiguous(), cs, [np.arange(64, dtype=np.int32)], stage_mask=0x0F)
    med = torch.median(cen, dim=1).values  # lower median for even G
    srt = torch.sort(cen, dim=1).values
    mid = (srt[:, G // 2 - 1] + srt[:, G // 2]) * 0.5
    assert mid.abs().max().item() < 1e-15
    # (4) the tumour clones' arm-level events are called: clone 0 gains chr1, loses chr10
    _, obs = synth.groups(C)
    c0 = int(obs[0][0])
    col = st[c0].cpu().numpy()
    assert (col[cs[0]:cs[1]] >= 4).mean() > 0.9 and (col[cs[9]:cs[10]] <= 2).mean() > 0.9
    # (5) checksum of checksums is reproducible run to run (deterministic reductions)
    out3, _ = dev.smooth_chain(x, cs, refs)
    assert torch.equal(out3, out)


# ------------------------------------------------------------------ ingest (steps 3-4) and the whole replay from counts
def test_ingest_and_full_replay_from_counts(dev, example):
    """@count.data -> steps 3,4 (normalise by depth, log2(x+1)) -> chain -> @expr.data, all on the HIP path."""
    from infercnv_amd import GeneOrder, InfercnvObject, ops
    counts = example["count_data"].astype(np.float64)
    levels = example["chr_levels"][example["chr_codes"]]
    obj = InfercnvObject(expr_data=counts, gene_order=GeneOrder(chr=levels),
                         reference_grouped_cell_indices={"normal": example["ref_normal"]},
                         observation_grouped_cell_indices={"tumor": example["obs_tumor"]})
    o3 = ops.normalize_counts_by_seq_depth(obj)
    want3 = onp.normalize_counts_by_seq_depth(counts)
    assert np.abs(o3.expr_data - want3).max() < 1e-9 * np.abs(want3).max()
    o4 = ops.log2xplus1(o3)
    assert np.abs(o4.expr_data - example["log"]).max() < 1e-12
    final = ops.hip_smooth_chain(o4)
    _, ref_pre, (mu, s) = oc.smooth_chain(example["log"], example["chr_start"], [example["ref_normal"]], want_pre_denoise=True)
    check_denoise_flips(final.expr_data, example["expr_data"], ref_pre, mu, s, tol=1e-11, expect=0, label="replay from counts")
    # device-resident flavour with an explicit factor
    xd = to_dev(counts)
    cs = dev.col_sums(xd).cpu().numpy()
    assert np.abs(cs - counts.sum(axis=0)).max() < 1e-6
    y = dev.normalize_log2(xd, normalize_factor=1e5)
    want, _ = oc.normalize_log2(counts, 1e5)
    assert np.abs(to_host(y) - want).max() < 1e-12


# ------------------------------------------------------------------ BASELINE.json configs 4 and 5 (scaled to one GPU)
def test_config4_i3_subclusters_properties(dev):
    """i3 HMM at subcluster level (500-cell subclusters): every member of a subcluster carries the subcluster's
    trace, and the trace equals the oracle's Viterbi on the GPU's own group means (identical inputs)."""
    from infercnv_amd import synth
    G, C = 10000, 20000
    x, cs = synth.make_matrix_torch(G, C, "cuda")
    refs, _ = synth.groups(C)
    out, pre = dev.smooth_chain(x, cs, refs, want_pre_denoise=True)
    ref_idx = np.concatenate(refs)
    mu, sigma = dev.cells_mean_sd(pre, ref_idx)
    delta_m = abs(-1.6448536269514722 * sigma)
    m3 = np.array([mu - delta_m, mu, mu + delta_m])
    Pi, dl = onp.get_HMM_i3(1e-6)
    groups = [np.arange(s, min(s + 500, C), dtype=np.int32) for s in range(0, C, 500)]
    st, bad = dev.viterbi_groups(pre, cs, groups, m3, [sigma] * len(groups), np.log(Pi), np.log(dl))
    assert int(bad.item()) == 0
    gm = dev.group_means(pre, groups)
    torch.cuda.synchronize()
    st_h = st.cpu().numpy()
    for q in (0, 7, len(groups) - 1):
        want, _ = oc.viterbi_cells(gm[q].cpu().numpy().reshape(-1, 1), cs, m3, sigma, np.log(Pi), np.log(dl))
        members = groups[q]
        assert (st_h[members] == want[:, 0][None, :]).all()
    assert set(np.unique(st_h)) <= {1, 2, 3}


def test_config5_median_filter_properties(dev):
    """2-D median denoise: constants are fixed points; the filter commutes with exact scalings and negation; tiles are
    independent (filtering a tile alone gives the same values)."""
    from infercnv_amd import synth
    G, C = 10000, 1500
    x, cs = synth.make_matrix_torch(G, C, "cuda")
    tiles = [np.arange(s, min(s + 500, C), dtype=np.int32) for s in range(0, C, 500)]
    y = dev.median_filter(x, cs, tiles, 7)
    assert torch.equal(dev.median_filter(x * 4.0, cs, tiles, 7), y * 4.0)      # power-of-two scaling is exact
    assert torch.equal(dev.median_filter(-x, cs, tiles, 7), -y)                # order statistics mirror exactly
    const = torch.full_like(x, 1.25)
    assert torch.equal(dev.median_filter(const, cs, tiles, 7), const)
    sub = x[500:1000].contiguous()
    ysub = dev.median_filter(sub, cs, [np.arange(500, dtype=np.int32)], 7)
    assert torch.equal(ysub, y[500:1000])
    # spot parity against the oracle on one chromosome of one tile
    k = 20                                                       # chr21: 108 genes
    xs = x[:500, cs[k]:cs[k + 1]].cpu().numpy().T
    want = oc.median_filter(xs, np.array([0, xs.shape[0]], dtype=np.int32), [np.arange(500, dtype=np.int32)], 7)
    np.testing.assert_array_equal(y[:500, cs[k]:cs[k + 1]].cpu().numpy().T, want)


# ------------------------------------------------------------------ state consensus + CNV region reports (8f #2)
def test_state_consensus_bit_exact(dev):
    rng = np.random.default_rng(21)
    G, C = 1500, 131
    st = rng.integers(1, 7, size=(G, C)).astype(np.uint8)
    st[rng.random((G, C)) < 0.05] = 255                           # the reference's -1 "untouched" entries
    st[:40, :] = np.where(rng.random((40, C)) < 0.5, 3, 4)        # many exact ties
    perm = rng.permutation(C)
    groups = [perm[:1], perm[1:3], perm[3:60], perm[60:124]]      # sizes 1, 2, odd, even; 7 cells in no group
    sf = st.astype(np.float64)
    sf[st == 255] = -1.0
    want = onp.state_consensus(sf, groups)
    d_st = torch.from_numpy(np.ascontiguousarray(st.T)).cuda()
    cons, over = dev.state_consensus(d_st, groups, overwrite=True)
    got = cons.cpu().numpy().T.astype(np.float64)
    got[got == 255] = -1
    assert np.array_equal(got, want)
    over = over.cpu().numpy().T
    for gi, g in enumerate(groups):
        assert np.array_equal(over[:, g], np.repeat(cons.cpu().numpy()[gi][:, None], len(g), axis=1))
    rest = perm[124:]
    assert np.array_equal(over[:, rest], st[:, rest])             # non-member cells keep their states


def test_cnv_region_reports_on_reference_states(dev, example, golden_dir, tmp_path):
    """get_predicted_CNV_regions / generate_cnv_region_reports on the reference's own HMM_states object
    (data/HMM_states.rda, mirrored in tests/golden/hmm_states_example.npz) against the loop restatement."""
    from infercnv_amd import cnv_regions
    from infercnv_amd.infercnv_object import GeneOrder, InfercnvObject
    hs = np.load(os.path.join(golden_dir, "hmm_states_example.npz"))["HMM_states"].astype(np.float64)
    chr_names = example["chr_levels"][example["chr_codes"] - example["chr_codes"].min()]
    obj = InfercnvObject(expr_data=hs, gene_order=GeneOrder(chr_names, example["gene_start"], example["gene_stop"]),
                         reference_grouped_cell_indices={"normal": example["ref_normal"]},
                         observation_grouped_cell_indices={"tumor": example["obs_tumor"]},
                         tumor_subclusters={"subclusters": {"normal": {"normal_s1": example["ref_normal"]},
                                                            "tumor": {"tumor_s1": example["obs_tumor"][:6],
                                                                      "tumor_s2": example["obs_tumor"][6:]}}})
    for by in ("consensus", "subcluster", "cell"):
        res = cnv_regions.get_predicted_CNV_regions(obj, by)
        counter = 0
        assert len(res) == {"consensus": 2, "subcluster": 3, "cell": 20}[by]
        for entry in res:
            idx = np.array([int(c.split("_")[1]) - 1 for c in entry["cells"]])
            cons = onp.state_consensus(hs, [idx])[:, 0]
            want, counter = onp.define_cnv_gene_regions(cons, list(chr_names), counter)
            assert [rn for rn, _ in entry["gene_regions"]] == [w[0] for w in want]
            for (rn, r), w in zip(entry["gene_regions"], want):
                assert r["state"] == w[1] and r["gene"].tolist() == w[2]
            for (rn, state, c, s, e), w in zip(entry["cnv_ranges"], want):
                assert s == example["gene_start"][w[2]].min() and e == example["gene_stop"][w[2]].max()
    assert res[0]["cell_group_name"] == "cell_%d" % (example["ref_normal"][0] + 1)
    regions = cnv_regions.generate_cnv_region_reports(obj, "HMM_pred", str(tmp_path), ignore_neutral_state=3,
                                                      by="subcluster")
    lines = open(tmp_path / "HMM_pred.pred_cnv_regions.dat").read().splitlines()
    assert lines[0] == "cell_group_name\tcnv_name\tstate\tchr\tstart\tend"
    n_non_neutral = sum(1 for x in regions for r in x["cnv_ranges"] if r[1] != 3)
    assert len(lines) == 1 + n_non_neutral and all(l.split("\t")[2] != "3" for l in lines[1:])
    assert lines[1].split("\t")[0] in ("normal.normal_s1", "tumor.tumor_s1", "tumor.tumor_s2")
    genes = open(tmp_path / "HMM_pred.pred_cnv_genes.dat").read().splitlines()
    assert genes[0] == "cell_group_name\tgene_region_name\tstate\tgene\tchr\tstart\tend"
    assert len(genes) == 1 + sum(len(r["gene"]) for x in regions for _, r in x["gene_regions"] if r["state"] != 3)
    assert open(tmp_path / "HMM_pred.cell_groupings").read().splitlines()[0] == "cell_group_name\tcell"
    used = open(tmp_path / "HMM_pred.genes_used.dat").read().splitlines()
    assert used[0] == "chr\tstart\tstop" and len(used) == 1 + hs.shape[0]
    # consensus overwrite (R/inferCNV_HMM.R:473-483): every subcluster becomes constant per gene
    new = cnv_regions.overwrite_with_consensus(obj, [example["obs_tumor"][:6], example["obs_tumor"][6:]])
    want = onp.state_consensus(hs, [example["obs_tumor"][:6]])[:, 0]
    assert np.array_equal(new.expr_data[:, example["obs_tumor"][:6]], np.repeat(want[:, None], 6, axis=1))
    assert np.array_equal(new.expr_data[:, example["ref_normal"]], hs[:, example["ref_normal"]])


def test_per_chr_subcluster_predictor_with_consensus(dev):
    """predict_CNV_via_HMM_on_tumor_subclusters_per_chr (R/inferCNV_HMM.R:412-487): per-chromosome
    subclusters, then the global subclusters' consensus overwrite."""
    from infercnv_amd import hmm, synth
    from infercnv_amd.infercnv_object import GeneOrder, InfercnvObject
    rng = np.random.default_rng(5)
    G, C = 900, 40
    chrs = np.array(["chr1"] * 400 + ["chr2"] * 300 + ["chr3"] * 200)
    x = 1.0 + 0.1 * rng.standard_normal((G, C))
    x[100:250, 20:] += 0.5
    x[450:600, :10] -= 0.45
    cms = {k: {"mean": m, "sd": 0.08} for k, m in zip(hmm.CNV_LEVELS, (0.01, 0.5, 1.0, 1.5, 2.0, 3.0))}
    sub = {"subclusters": {"all": {"s1": np.arange(0, 20), "s2": np.arange(20, 40)}}}
    per_chr = {"chr1": [np.arange(0, 20), np.arange(20, 40)],
               "chr2": [np.arange(0, 10), np.arange(10, 25), np.arange(25, 40)],
               "chr3": [np.arange(0, 40)]}
    obj = InfercnvObject(expr_data=x, gene_order=GeneOrder(chrs),
                         observation_grouped_cell_indices={"all": np.arange(C)}, tumor_subclusters=sub)
    got = hmm.predict_CNV_via_HMM_on_tumor_subclusters_per_chr(obj, per_chr, cms).expr_data
    # oracle: Viterbi of each (chr, subcluster) mean profile (the GPU's own group means -> identical inputs),
    # then the per-gene consensus of each global subcluster
    h = hmm._get_HMM(cms, 1e-6)
    means, logPi, logDelta = h["state_emission_params"]["mean"], np.log(h["state_transitions"]), np.log(h["delta"])
    raw = np.full((G, C), -1.0)
    for c, groups in per_chr.items():
        rows = np.nonzero(chrs == c)[0]
        gm = dev.group_means(to_dev(x[rows]), groups).cpu().numpy().T
        assert np.abs(gm - oc.group_means(np.ascontiguousarray(x[rows]), groups)).max() < 1e-14
        for q, g in enumerate(groups):
            st, _ = oc.viterbi_cells(gm[:, q:q + 1], np.array([0, rows.size], dtype=np.int32), means, 0.08, logPi, logDelta)
            raw[np.ix_(rows, g)] = st.astype(np.float64)
    want = raw.copy()
    for g in (np.arange(0, 20), np.arange(20, 40)):
        want[:, g] = onp.state_consensus(raw, [g])
    assert np.array_equal(got, want)
    assert (got[100:250, 20:] == 4).mean() > 0.9 and (got[450:600, :10] == 2).mean() > 0.0


# ------------------------------------------------------------------ gene filters + spike-in statistics (8f #1, #3)
def test_gene_filters_and_row_selection(dev, example):
    """Step 2 on the reference's bundled raw counts: require_above_min_mean_expr_cutoff / require_above_min_cells_ref
    decide exactly as R does (integer counts: the fp64 row sums are exact), remove_genes gathers the rows."""
    from infercnv_amd import ops
    from infercnv_amd.infercnv_object import GeneOrder, InfercnvObject
    counts = example["count_data"].astype(np.float64)
    rng = np.random.default_rng(2)
    counts[rng.random(counts.shape[0]) < 0.2] *= 0.0                  # silence some genes entirely
    counts[rng.random(counts.shape) < 0.5] = 0.0
    obj = InfercnvObject(expr_data=counts, count_data=counts.copy(), gene_order=GeneOrder(example["chr_codes"], example["gene_start"],
                                                                                       example["gene_stop"]),
                         reference_grouped_cell_indices={"normal": example["ref_normal"]},
                         observation_grouped_cell_indices={"tumor": example["obs_tumor"]})
    sums, nnz = ops._gene_stats(obj)
    assert np.array_equal(sums, counts.sum(axis=1)) and np.array_equal(nnz, (counts > 0).sum(axis=1))
    for cutoff in (0.1, 1.0, 2.5):
        got = ops.require_above_min_mean_expr_cutoff(obj, cutoff)
        drop = onp.below_min_mean_expr_cutoff(counts, cutoff)
        keep = np.setdiff1d(np.arange(counts.shape[0]), drop)
        assert 0 < drop.size < counts.shape[0]
        assert np.array_equal(got.expr_data, counts[keep]) and np.array_equal(got.gene_order.start, example["gene_start"][keep])
    got = ops.require_above_min_cells_ref(obj, 3)
    assert np.array_equal(got.expr_data, counts[onp.genes_passing_min_cells(counts, 3)])
    with pytest.raises(RuntimeError):
        ops.require_above_min_cells_ref(obj, 10 ** 6)
    # device-resident flavour on a larger matrix, NaN entries are "not expressed"
    x = rng.poisson(0.7, size=(5000, 700)).astype(np.float64)
    x[rng.random(x.shape) < 0.01] = np.nan
    L = __import__("infercnv_amd")._lib.load()
    import ctypes as ct
    d = to_dev(x)
    s_d = torch.empty(5000, dtype=torch.float64, device="cuda")
    n_d = torch.empty(5000, dtype=torch.int32, device="cuda")
    rc = L.icnv_gene_stats_dev(ct.c_void_p(d.data_ptr()), 5000, 700, ct.c_void_p(s_d.data_ptr()), ct.c_void_p(n_d.data_ptr()), None)
    assert rc == 0
    assert np.array_equal(n_d.cpu().numpy(), onp.genes_passing_min_cells(x, 0).size * 0 + (x > 0).sum(axis=1))
    ok = ~np.isnan(x).any(axis=1)
    assert np.abs(s_d.cpu().numpy()[ok] - x[ok].sum(axis=1)).max() == 0.0


def test_get_spike_dists_block_statistics(dev):
    """get_spike_dists (R/inferCNV_HMM.R:15-99) on a synthetic hidden spike-in object."""
    from infercnv_amd import hmm
    from infercnv_amd.infercnv_object import GeneOrder, InfercnvObject
    rng = np.random.default_rng(8)
    sizes = [40, 35, 40, 30, 40, 45, 40, 25, 40, 50, 90]
    chrs = np.concatenate([[name] * n for (name, _), n in zip(hmm.HSPIKE_CHR_INFO, sizes)])
    level = np.concatenate([[cnv] * n for (_, cnv), n in zip(hmm.HSPIKE_CHR_INFO, sizes)])
    C = 60
    x = rng.normal(1.0, 0.1, size=(chrs.size, C))
    x[:, 30:] *= level[:, None]                                        # spiked cells
    obj = InfercnvObject(expr_data=x, gene_order=GeneOrder(chrs),
                         reference_grouped_cell_indices={"simnormal": np.arange(0, 30)},
                         observation_grouped_cell_indices={"spike_a": np.arange(30, 50), "spike_b": np.arange(50, 60)})
    got = hmm.get_spike_dists(obj)
    assert list(got) == ["cnv:1", "cnv:0.01", "cnv:0.5", "cnv:1.5", "cnv:2", "cnv:3"]
    for key, v in got.items():
        cnv = float(key.split(":")[1])
        m, s = onp.gene_expr_mean_sd(x, np.nonzero(level == cnv)[0], np.arange(30, 60))
        assert abs(v["mean"] - m) < 1e-14 and abs(v["sd"] - s) < 1e-14
    # the result feeds the i6 predictor directly
    st = hmm.predict_CNV_via_HMM_on_indiv_cells(obj, got).expr_data
    assert (st[level == 3][:, 30:] == 6).mean() > 0.9 and (st[level == 1][:, :30] == 3).mean() > 0.9


# ------------------------------------------------------------------ cell-cell distances
@pytest.mark.parametrize("G,C,n", [(2000, 300, 150), (2001, 90, 70), (512, 64, 64), (777, 200, 129), (64, 5, 1), (33, 40, 2)])
def test_cell_distances_vs_oracle(dev, G, C, n):
    """parallelDist(t(expr[, cells])) (R/inferCNV_tumor_subclusters.R:191): the fp64 MFMA Gram formulation against
    direct sums of squared differences; relative 1e-12 (tile edges, odd G, non-multiples of the 64-cell tile)."""
    rng = np.random.default_rng(G + n)
    x = rng.normal(1.0, 0.3, size=(G, C)) + rng.normal(0.0, 2.0, size=(G, 1))     # large per-gene offsets: centring matters
    cells = rng.permutation(C)[:n].astype(np.int32)
    got = dev.cell_distances(to_dev(x), cells).cpu().numpy()
    want = onp.cell_distances(x, cells)
    assert got.shape == (n, n) and np.all(np.diag(got) == 0.0)
    np.testing.assert_array_equal(got, got.T)
    assert np.abs(got - want).max() <= 1e-12 * max(want.max(), 1.0)
    # translation invariance per gene (the kernel centres internally): exact same structure, tiny numeric change
    got2 = dev.cell_distances(to_dev(x + 5.0), cells).cpu().numpy()
    assert np.abs(got2 - want).max() <= 1e-11 * max(want.max(), 1.0)
    # duplicate cells are at distance (numerically) zero
    if n >= 2:
        dup = np.concatenate([cells[:2], cells[:1]]).astype(np.int32)
        d3 = dev.cell_distances(to_dev(x), dup).cpu().numpy()
        assert d3[0, 2] <= 1e-6 * max(want.max(), 1.0) and abs(d3[0, 1] - want[0, 1]) <= 1e-12 * max(want.max(), 1.0)


def test_parallelDist_mirror_matches_scipy(dev):
    """Host mirror of parallelDist(t(tumor_expr_data)): the R `dist` vector (== SciPy's condensed form)."""
    from scipy.spatial.distance import pdist
    from infercnv_amd import GeneOrder, InfercnvObject, tumor_subclusters
    rng = np.random.default_rng(8)
    G, C = 900, 70
    x = rng.normal(1.0, 0.2, size=(G, C))
    obj = InfercnvObject(expr_data=x, gene_order=GeneOrder(chr=np.repeat(["chr1", "chr2"], [400, 500])),
                         reference_grouped_cell_indices={"normal": np.arange(10)},
                         observation_grouped_cell_indices={"tumor": np.arange(10, C)})
    cells = np.arange(10, C)
    d = tumor_subclusters.parallelDist(obj, cells)
    want = pdist(x[:, cells].T)
    assert d.shape == want.shape and np.abs(d - want).max() <= 1e-12 * want.max()


def test_chain_missing_values_policy_against_the_reference_semantics(dev):
    """What happens to NA / NaN in the chain, stated against the reference.  R: `.smooth_helper` strips a cell's NAs,
    smooths the shortened sequence and re-inserts them (R/inferCNV_ops.R:2487-2489, 2529); `median(x, na.rm = TRUE)`
    centres on the values present (:2098); the NA itself stays NA through every step.  run() cannot produce one (its
    chain input is log2(x + 1) of counts), and the library does NOT restate that path.  Its policy, asserted here:
      * a cell WITHOUT a NaN is untouched by NaNs elsewhere in the matrix (same values as the oracle on the clean data),
        as long as the NaN is not in a reference cell (there it makes that gene's reference mean NaN, as rowMeans does);
      * inside the affected cell, stand-alone smoothing spreads the NaN over its half window on that chromosome
        (R: only the NA position stays NA, its neighbours are smoothed over the gap) -- every other chromosome of the cell
        equals the reference's result;
      * with step 9 in the chain the NaN is clamped to the threshold (v_min / v_max return the number): the cell comes out
        finite where R keeps an NA."""
    from infercnv_amd import synth
    G, C = 1500, 24
    x, cs = synth.make_matrix_np(G, C)
    x = x - 1.5
    refs = [np.arange(0, 4, dtype=np.int32)]
    bad_cell, bad_gene = 9, 400
    xn = x.copy()
    xn[bad_gene, bad_cell] = np.nan
    chr_codes = np.repeat(np.arange(len(cs) - 1), np.diff(cs))
    k = int(chr_codes[bad_gene])
    # stand-alone smoothing (step 10)
    got = to_host(dev.smooth_chain(to_dev(xn), cs, refs, stage_mask=0x04)[0])
    clean = np.delete(np.arange(C), bad_cell)
    want_clean = oc.smooth_by_chromosome(x, cs, 101)
    assert np.abs(got[:, clean] - want_clean[:, clean]).max() < 1e-11
    r_col = np.concatenate([onp.smooth_window_na(xn[cs[j]:cs[j + 1], bad_cell], 101) for j in range(len(cs) - 1)])
    other = chr_codes != k
    assert np.abs(got[other, bad_cell] - r_col[other]).max() < 1e-11            # other chromosomes of the cell: the reference's values
    assert np.isnan(r_col).sum() == 1 and np.isnan(r_col[bad_gene])             # R: exactly the NA position stays NA
    lib_nan = np.nonzero(np.isnan(got[:, bad_cell]))[0]
    assert lib_nan.min() >= max(cs[k], bad_gene - 50) and lib_nan.max() <= min(cs[k + 1] - 1, bad_gene + 50) and bad_gene in lib_nan
    # stand-alone median centring (step 11): R centres the cell on its present values and keeps the NA
    got11 = to_host(dev.smooth_chain(to_dev(xn), cs, refs, stage_mask=0x08)[0])
    assert np.abs(got11[:, clean] - onp.center_columns_na(xn)[:, clean]).max() == 0.0
    assert np.isfinite(got11[np.arange(G) != bad_gene, bad_cell]).all()          # the cell terminates and stays finite elsewhere
    # the full chain: step 9 clamps the NaN to the threshold -> finite output where R has an NA
    out, pre = dev.smooth_chain(to_dev(xn), cs, refs, want_pre_denoise=True)
    assert np.isfinite(to_host(pre)).all() and np.isfinite(to_host(out)).all()
    want_pre = oc.smooth_chain(x, cs, refs, want_pre_denoise=True)[1]
    assert np.abs(to_host(pre)[:, clean] - want_pre[:, clean]).max() < 1e-11


@pytest.mark.parametrize("case", ["full_chain", "no_bounds", "smooth_only", "centre_median", "centre_mean", "nan_in_reference_cell",
                                  "full_chain_in_place", "no_bounds_in_place", "no_bounds_nan_in_reference_cell",
                                  "from_step_9_nan_in_reference_cell",
                                  "full_chain:two_pass", "nan_in_reference_cell:two_pass", "no_bounds:three_pass_taps"])
def test_chain_na_aware_reference_semantics(dev, case, monkeypatch):
    """ICNV_ST_NA_AWARE (round 5): the cells that hold a NaN come out the way the reference's step functions treat an NA
    (oracle_np.run_chain_na: R/inferCNV_ops.R:1757-1768 which() never selects an NA -> 0 with bounds; :2974-2975 the clamp
    leaves it alone; :2487-2489, 2529 the smoothing strips and re-inserts NAs per chromosome; :2098 median(na.rm = TRUE);
    2^NA = NA; :2335 step 22 never selects it), every other cell as always.  NaNs at chromosome starts and ends, runs of them,
    a whole chromosome of one cell, a cell with a single value left on a chromosome, in observation and in reference cells."""
    from infercnv_amd import synth, _lib
    if ":" in case:
        # round 6: the NA pass behind the chain for gene sets beyond the LDS-resident limit (forced on this size): two-pass form
        # (strided views + centre / finish) and the (2T + 1)-tap three-pass chain
        case, form = case.split(":")
        monkeypatch.setenv("ICNV_CHAIN_LARGE", "1")
        if form == "three_pass_taps":
            monkeypatch.setenv("ICNV_CHAIN_LARGE_TAPS", "1")
    G, C = 1500, 40
    x, cs = synth.make_matrix_np(G, C)
    x = x - 1.5
    refs = [np.arange(0, 4, dtype=np.int32), np.arange(4, 9, dtype=np.int32)]
    chr_codes = np.repeat(np.arange(len(cs) - 1), np.diff(cs))
    xn = x.copy()
    rng = np.random.default_rng(len(case))
    xn[400, 12] = np.nan                                       # a single NA
    xn[cs[3], 15] = np.nan; xn[cs[4] - 1, 15] = np.nan         # first and last gene of a chromosome
    xn[600:640, 17] = np.nan                                   # a run
    xn[cs[5]:cs[6], 20] = np.nan                               # a whole chromosome of one cell
    xn[cs[7]:cs[8] - 1, 21] = np.nan                           # one value left on a chromosome
    xn[rng.integers(0, G, 60), 25] = np.nan                    # scattered
    xn[:, 30] = np.nan                                         # a cell of nothing but NAs
    if "nan_in_reference_cell" in case:
        xn[700, 2] = np.nan                                    # the gene's reference mean is NA: with bounds the whole gene comes out 0,
        xn[900:905, 6] = np.nan                                # without bounds x - NA = NA for that gene in EVERY cell
    na = _lib.ST_NA_AWARE
    xd = to_dev(xn)
    if case == "from_step_9_nan_in_reference_cell":
        # round 6 (ADVICE): a chain WITHOUT step 8 leaves a reference cell's NaN in its cached column; the reference cells' share of
        # the apply continues from the cache and must treat it the reference's way too (step 12 with bounds: NA -> 0)
        m = 0x7F & ~0x01
        out, pre = dev.smooth_chain(xd, cs, refs, stage_mask=m | na, want_pre_denoise=True)
        with np.errstate(invalid="ignore"):                    # the step functions of oracle_np.run_chain_na from step 9 on
            y = onp.apply_max_threshold_bounds(xn, 3.0)
            y = onp.smooth_by_chromosome_na(y, chr_codes, 101)
            y = onp.center_columns_na(y)
            y = onp.subtract_expr(y, onp.get_normal_gene_mean_bounds(y, refs), True)
            want_pre = onp.invert_log2(y)
        got_pre = to_host(pre)
        assert not np.isnan(want_pre).any()                    # step 12 with bounds turns every NA (value or bound) into 0 ...
        assert (want_pre[700] == 1.0).all() and (want_pre[900:905] == 1.0).all()   # ... the genes whose reference mean is NA: 2^0
        assert not np.isnan(got_pre).any()
        assert np.abs(got_pre - want_pre).max() < 1e-11
        return
    if case in ("full_chain", "no_bounds", "nan_in_reference_cell", "full_chain_in_place", "no_bounds_in_place", "no_bounds_nan_in_reference_cell"):
        ub = not case.startswith("no_bounds")
        if case.endswith("_in_place"):
            # round 6 (ADVICE, medium): expr_out may alias expr_in (include/icnv.h); the NA pass must still see the ORIGINAL columns
            work = xd.clone()
            out, pre = dev.smooth_chain(work, cs, refs, use_bounds=ub, stage_mask=0x7F | na, want_pre_denoise=True, out=work)
            assert out.data_ptr() == work.data_ptr()
        else:
            out, pre = dev.smooth_chain(xd, cs, refs, use_bounds=ub, stage_mask=0x7F | na, want_pre_denoise=True)
        want, want_pre = onp.run_chain_na(xn, chr_codes, refs, use_bounds=ub, return_pre_denoise=True)
        got_pre, got = to_host(pre), to_host(out)
        if case == "no_bounds_nan_in_reference_cell":
            assert np.isnan(want_pre[700]).all() and np.isnan(want_pre[900:905]).all()   # x - NA: the whole gene, every cell
        assert np.array_equal(np.isnan(got_pre), np.isnan(want_pre)), case
        if ub:
            assert not np.isnan(want_pre[:, :30]).any() and np.isnan(want_pre[:, 30]).sum() == 0   # with bounds step 8 turns every NA into 0
        ok = ~np.isnan(want_pre)
        assert np.abs(got_pre[ok] - want_pre[ok]).max() < 1e-11, case
        if not np.isnan(want).all():
            mu, s = onp.clear_noise_params_via_ref_mean_sd(want_pre, np.concatenate(refs), 1.5)
            if np.isfinite(mu) and np.isfinite(s):
                fin = ~np.isnan(want).any(axis=0)
                check_denoise_flips(got[:, fin], want[:, fin], want_pre[:, fin], mu, s, tol=1e-11, label=f"NA-aware {case}")
            assert np.array_equal(np.isnan(got), np.isnan(want))
    elif case == "smooth_only":
        got = to_host(dev.smooth_chain(xd, cs, refs, stage_mask=0x04 | na)[0])
        want = onp.smooth_by_chromosome_na(xn, chr_codes, 101)
        assert np.array_equal(np.isnan(got), np.isnan(xn))                      # exactly the NA positions stay NA
        ok = ~np.isnan(want)
        assert np.abs(got[ok] - want[ok]).max() < 1e-11
    else:
        mask = 0x08 if case == "centre_median" else 0x88
        got = to_host(dev.smooth_chain(xd, cs, refs, stage_mask=mask | na)[0])
        with np.errstate(invalid="ignore"), __import__("warnings").catch_warnings():
            __import__("warnings").simplefilter("ignore")
            want = onp.center_columns_na(xn) if case == "centre_median" else None
        if case == "centre_median":
            ok = ~np.isnan(want)
            assert np.array_equal(np.isnan(got), np.isnan(xn))
            assert np.abs(got[ok] - want[ok]).max() == 0.0                      # the median of the values present, exactly
        else:
            with np.errstate(invalid="ignore"), __import__("warnings").catch_warnings():
                __import__("warnings").simplefilter("ignore")
                m = np.nanmean(xn, axis=0)
            want = xn - m[None, :]
            ok = ~np.isnan(want)
            assert np.array_equal(np.isnan(got), np.isnan(xn))
            assert np.abs(got[ok] - want[ok]).max() < 1e-12
    # without the flag nothing changes for the cells that hold no NaN (they never went through the slow path)
    clean = [c for c in range(C) if not np.isnan(xn[:, c]).any()]
    if case == "full_chain":
        _, pre0 = dev.smooth_chain(to_dev(x), cs, refs, stage_mask=0x7F, want_pre_denoise=True)
        assert np.abs(to_host(pre)[:, clean] - to_host(pre0)[:, clean]).max() < 1e-11


@pytest.mark.parametrize("G,kw", [(10000, {}), (10240, {}), (6002, {}), (10000, {"stage_mask": 0x7F & ~0x20}), (10000, {"stage_mask": 0x7F & ~0x10})])
def test_denoise_round_streaming_kernel_equals_chain_geometry(dev, G, kw, monkeypatch):
    """The step-22 parameters (mean of the reference values, mean of the reference cells' sds) come from a streaming kernel over
    the reference-cell cache when the remaining stages are elementwise (even G <= 10 240), from the chain geometry otherwise.
    The two add in different orders: forced through the chain geometry (ICNV_CELL_STATS_CHAIN=1) the parameters must agree to
    rounding, for the bench's gene count, the largest gene count the streaming kernel takes, a small even one, and chains
    without step 14 / without step 12."""
    from infercnv_amd import synth
    C = 700
    x, cs = synth.make_matrix_np(G, C)
    refs, _ = synth.groups(C, ref_frac=0.2)
    xd = to_dev(x)
    def params():
        plan = dev.ChainPlan(G, C, cs, refs, **kw)
        for r in range(plan.num_rounds):
            plan.round_partial(r, xd); plan.round_finish(r)
        mu_s = plan.denoise_params()
        plan.close()
        return mu_s
    monkeypatch.delenv("ICNV_CELL_STATS_CHAIN", raising=False)
    a = params()
    monkeypatch.setenv("ICNV_CELL_STATS_CHAIN", "1")
    b = params()
    assert abs(a[0] - b[0]) <= 1e-13 * abs(b[0]) and abs(a[1] - b[1]) <= 1e-12 * abs(b[1]), (a, b)
    want = oc.smooth_chain(x, cs, refs, want_pre_denoise=True, **kw)[2]
    assert abs(a[0] - want[0]) <= 1e-12 and abs(a[1] - want[1]) <= 1e-12, (a, want)


# ------------------------------------------------------------------ ragged and degenerate layouts (round 4)
@pytest.mark.parametrize("sizes,C", [
    ((1, 2, 15, 16, 17, 31, 33, 1, 129, 255, 3), 200),      # single-gene chromosomes (state 3, R/inferCNV_HMM.R:1104-1107), lengths around the 16-gene block
    ((16,) * 9 + (1,), 65),                                  # every chromosome exactly one block; one column more than a wavefront
    ((1072, 7, 708), 64),                                    # a whole wavefront of columns, chromosome starts at odd offsets
    ((640, 641), 127),                                       # G odd: state columns differ in their alignment from lane to lane (no block summaries)
])
def test_viterbi_ragged_layouts_bit_exact(dev, sizes, C):
    """The certified fast path on layouts that exercise its alignment logic -- 128-byte observation chunks, the per-gene
    head / tail of a sequence, 16-gene block summaries of the traceback (used only when the states' 16-byte alignment is
    wave-uniform and goes together with the observations' line alignment), partial column blocks -- against the exact
    kernel and the oracle, i6 and i3."""
    from infercnv_amd import synth
    G = int(sum(sizes))
    cs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    rng = np.random.default_rng(G + C)
    means, sd, logPi, logDelta = synth.hmm_params_i6()
    level = rng.choice(means, size=(len(sizes), C))                       # one CNV level per (chromosome, cell) ...
    x = np.repeat(level, sizes, axis=0) + rng.normal(0.0, 0.08, size=(G, C))
    x[rng.integers(0, G, 40), rng.integers(0, C, 40)] += rng.choice([-0.4, 0.4], 40)   # ... and a few excursions
    xd = to_dev(x)
    st, bad = dev.viterbi_cells(xd, cs, means, sd, logPi, logDelta)
    assert dev.viterbi_last_stats()["path"] == ("fast" if C >= 64 else "exact")
    want, _ = oc.viterbi_cells(x, cs, means, sd, logPi, logDelta)
    np.testing.assert_array_equal(to_host(st), want)
    for k, n in enumerate(sizes):
        if n == 1:
            assert (want[cs[k]] == 3).all()
    dev.viterbi_set_mode(1)
    st_exact, _ = dev.viterbi_cells(xd, cs, means, sd, logPi, logDelta)
    dev.viterbi_set_mode(0)
    assert torch.equal(st, st_exact)
    # i3 on the same layout
    Pi, dl = onp.get_HMM_i3(1e-6)
    m3 = np.array([0.8, 1.0, 1.2])
    st3, _ = dev.viterbi_cells(xd, cs, m3, 0.1, np.log(Pi), np.log(dl))
    want3, _ = oc.viterbi_cells(x, cs, m3, 0.1, np.log(Pi), np.log(dl))
    np.testing.assert_array_equal(to_host(st3), want3)


@pytest.mark.parametrize("G", [3000, 4096])     # 3000: the columns' alignment alternates from lane to lane; 4096: wave-uniform
def test_viterbi_unaligned_matrix_views_bit_exact(dev, G):
    """The fast path on matrices that do not start on a cache line (a view into a larger allocation: observations 8 or 24
    bytes off a line, states 1 or 3 bytes off a 16-byte word): the chunk loop's alignment assumptions are wave-uniform
    decisions, not preconditions."""
    from infercnv_amd import synth
    pre, cs = _hmm_input(G, 130, seed=5)
    means, sd, logPi, logDelta = synth.hmm_params_i6()
    want, _ = oc.viterbi_cells(pre, cs, means, sd, logPi, logDelta)
    C = pre.shape[1]
    for off_x, off_s in ((1, 1), (3, 3), (0, 5), (2, 0)):
        bufx = torch.zeros(C * G + 8, dtype=torch.float64, device="cuda")
        xv = bufx[off_x:off_x + C * G].view(C, G)
        xv.copy_(to_dev(pre))
        bufs = torch.zeros(C * G + 16, dtype=torch.uint8, device="cuda")
        sv = bufs[off_s:off_s + C * G].view(C, G)
        st, bad = dev.viterbi_cells(xv, cs, means, sd, logPi, logDelta, states=sv)
        np.testing.assert_array_equal(to_host(st), want)
        assert int(bufs[:off_s].sum()) == 0 and int(bufs[off_s + C * G:].sum()) == 0     # nothing written outside the view


@pytest.mark.parametrize("G,C", [(9939, 700), (1003, 130), (4613, 64), (10000, 200)])
def test_padded_leading_dimension_chain_and_viterbi(dev, G, C):
    """Round 6: matrices with a leading dimension (device.padded_matrix: every cell's column starts on a cache line / a 16-byte word
    of its own).  The fused chain writes its HMM input into one (icnv_chain_apply_ld_dev) -- bit for bit the contiguous one --, the
    per-cell Viterbi reads it and writes padded states (icnv_viterbi_cells_ld_dev): the very states of the contiguous call and of
    the oracle, for gene counts that are not multiples of 16 (9 939 = example/run.R's, 4 613 = the golden object's) and one that
    is; the certified fast path serves both layouts; the padding columns are never touched."""
    from infercnv_amd import synth
    x, cs = synth.make_matrix_np(G, C)
    refs, _ = synth.groups(C)
    xd = to_dev(x)
    out_c, pre_c = dev.smooth_chain(xd, cs, refs, want_pre_denoise=True)
    plan = dev.ChainPlan(G, C, cs, refs)
    for r in range(plan.num_rounds):
        plan.round_partial(r, xd); plan.round_finish(r)
    pre_p = dev.padded_matrix(C, G)
    ld = pre_p.stride(0)
    assert ld % 16 == 0 and ld >= G and (ld > G) == (G % 16 != 0)
    base = pre_p.as_strided((C, ld), (ld, 1)) if ld > G else None       # the whole allocation, padding included
    if base is not None:
        base.fill_(-777.0)
    out_p, pre_p2 = plan.apply(xd, pre=pre_p)
    torch.cuda.synchronize()
    assert pre_p2.data_ptr() == pre_p.data_ptr()
    assert torch.equal(pre_p, pre_c) and torch.equal(out_p, out_c)
    if base is not None:
        assert bool((base[:, G:] == -777.0).all())                       # the padding columns are not written
    means, sd, logPi, logDelta = synth.hmm_params_i6()
    st_c, bad_c = dev.viterbi_cells(pre_c, cs, means, sd, logPi, logDelta)
    path_c = dev.viterbi_last_stats()
    st_p, bad_p = dev.viterbi_cells(pre_p, cs, means, sd, logPi, logDelta)
    path_p = dev.viterbi_last_stats()
    torch.cuda.synchronize()
    assert int(bad_c.item()) == 0 and int(bad_p.item()) == 0
    assert st_p.stride(0) == ld and st_p.shape == (C, G)
    assert torch.equal(st_p, st_c)
    assert path_c["path"] == path_p["path"] == "fast"
    want, _ = oc.viterbi_cells(to_host(pre_c), cs, means, sd, logPi, logDelta)
    np.testing.assert_array_equal(to_host(st_p.contiguous()), want)
    # i3 on the padded layout, and an error for a leading dimension below G
    mu, sigma, delta = onp.i3_params(to_host(pre_c), np.concatenate(refs), 0.05)
    Pi3, d3 = onp.get_HMM_i3(1e-6)
    m3 = np.array([mu - delta, mu, mu + delta])
    s3p, _ = dev.viterbi_cells(pre_p, cs, m3, sigma, np.log(Pi3), np.log(d3))
    s3c, _ = dev.viterbi_cells(pre_c, cs, m3, sigma, np.log(Pi3), np.log(d3))
    assert torch.equal(s3p, s3c)
    import ctypes as ct
    from infercnv_amd import _lib
    L = _lib.load()
    assert L.icnv_viterbi_cells_ld_dev(ct.c_void_p(pre_c.data_ptr()), G - 1, ct.c_void_p(st_c.data_ptr()), G, G, C, None, 0, 6, None, 0.1, None, None,
                                       None, None) == 1


def test_group_hmm_plan_i3_device_resident_parameters(dev):
    """Round 6: the i3 HMM at group level as a plan (icnv_group_hmm_*): group means and the reference cells' shifted moments in ONE
    pass, mu / sigma / delta derived on the device, the Viterbi reading them from device memory.  Against the call-by-call path
    (two-pass long-double moments on the host, icnv_viterbi_groups_dev) and the oracle: mu and sigma to 1e-13 relative, every
    state call identical; a second step on the same plan (nothing is uploaded again) and a KS-style explicit delta; groups the
    plan does not take (a cell in two groups) are refused with ICNV_ERR_UNSUPPORTED -- sharded.ShardedGroupHMM falls back."""
    import statistics
    from infercnv_amd import synth, sharded, _lib
    G, C = 3000, 900
    x, cs = synth.make_matrix_np(G, C)
    refs, _ = synth.groups(C)
    _, pre, _ = oc.smooth_chain(x, cs, refs, want_pre_denoise=True)
    subs, is_ref, _ = synth.subclusters(C)
    groups = [np.asarray(g, dtype=np.int32) for g in subs]
    ref_cells = np.concatenate([g for g, r in zip(groups, is_ref) if r])
    pd = to_dev(pre)
    mu0, sigma0 = dev.cells_mean_sd(pd, ref_cells)
    z = abs(statistics.NormalDist().inv_cdf(0.05))
    Pi3, d3 = onp.get_HMM_i3(1e-6)
    plan = dev.GroupHMMPlan(G, C, cs, groups, ref_cells)
    for step in range(2):
        m3 = plan.i3_partial(pd)
        st = plan.i3_finish(np.log(Pi3), np.log(d3), z, device=pd.device)
        mu, sigma, delta = plan.i3_params()
        assert abs(mu - mu0) <= 1e-13 * abs(mu0) and abs(sigma - sigma0) <= 1e-13 * sigma0, (mu, mu0, sigma, sigma0)
        assert abs(delta - sigma * z) <= 1e-15
        assert float(m3[2]) == ref_cells.size * G
        want, _ = dev.viterbi_groups(pd, cs, groups, np.array([mu0 - sigma0 * z, mu0, mu0 + sigma0 * z]), [sigma0] * len(groups), np.log(Pi3), np.log(d3))
        assert torch.equal(st, want), int((st != want).sum())
    ref_states, _ = oc.viterbi_groups(pre, cs, groups, np.array([mu0 - sigma0 * z, mu0, mu0 + sigma0 * z]), [sigma0] * len(groups), np.log(Pi3), np.log(d3))
    np.testing.assert_array_equal(to_host(st), ref_states)
    assert len(np.unique(ref_states)) >= 2
    # an explicit delta (the KS-based one of use_KS = TRUE is computed by the caller from sigma)
    st_k = plan.i3_finish(np.log(Pi3), np.log(d3), z, delta_abs=0.07, device=pd.device)
    want_k, _ = dev.viterbi_groups(pd, cs, groups, np.array([mu0 - 0.07, mu0, mu0 + 0.07]), [sigma0] * len(groups), np.log(Pi3), np.log(d3))
    assert torch.equal(st_k, want_k)
    plan.close()
    # the sharded driver takes the plan by itself and gives the call-by-call result
    hmm = sharded.ShardedGroupHMM()
    got = hmm.run_i3(pd, cs, groups, ref_cells)
    assert hmm._plan is not None and torch.equal(got, want)
    with pytest.raises(_lib.IcnvError) as e:
        dev.GroupHMMPlan(G, C, cs, groups + [groups[0][:3]], ref_cells)
    assert e.value.code == 3
    hmm2 = sharded.ShardedGroupHMM()
    got2 = hmm2.run_i3(pd, cs, groups + [groups[0][:3]], ref_cells)       # falls back to the call-by-call path
    assert hmm2._plan is None and got2.shape == got.shape
