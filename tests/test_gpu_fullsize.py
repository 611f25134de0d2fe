"""Full-size parity (-m gpu): BASELINE.json configs 2, 4 and 5 at 10 000 genes x 50 000 cells per GPU, the HIP
path through the C ABI against the CPU ORACLE over EVERY cell -- not against itself, not on a sample.

The oracle (oracle/icnv_oracle.c, OpenMP over cells) does the whole 10 000 x 50 000 chain + Viterbi in well under a
minute on the GPU box's host cores.  Cells are independent given the reference cells' statistics
(R/inferCNV_ops.R:1678-1786, 2302-2346), so the oracle runs over slices `[all reference cells | a block of
observation cells]`: every slice sees the same reference statistics as the whole matrix (asserted: the oracle's
step-22 parameters are identical from slice to slice), and the host never holds more than a few GB.

Tolerances: chain |delta| <= 1e-11 absolute on the pre-denoise matrix (north star: 1e-5 relative); step 22's strict
select accounted for element by element (tests/parity_util.py); HMM states bit-exact -- on identical inputs (the
oracle's Viterbi on the GPU's own chain output) AND end to end (the oracle's Viterbi on the oracle's own chain
output); median filter exact.
Reference semantics: R/inferCNV_ops.R:2335, R/inferCNV_HMM.R:1101-1176, 383, R/inferCNV_i3HMM.R:99-156, 249-308,
R/noise_reduction.R:92-113.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import oracle_c as oc  # noqa: E402
import oracle_np as onp  # noqa: E402
from parity_util import check_denoise_flips_t  # noqa: E402

G, C = 10000, 50000
BLOCK = 10000            # observation cells per oracle slice: a multiple of 2 000 = one 500-cell subcluster of each clone


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from infercnv_amd import device
    torch.cuda.set_device(0)
    device.init(0)
    device.viterbi_set_mode(0)
    return device


@pytest.fixture(scope="module")
def run50k(dev):
    """The HIP path on BASELINE config 2's matrix: chain (denoised + HMM input) and per-cell i6 states, kept in HBM."""
    from infercnv_amd import synth
    x, cs = synth.make_matrix_torch(G, C, "cuda")
    refs, _ = synth.groups(C)
    out, pre = dev.smooth_chain(x, cs, refs, want_pre_denoise=True)
    hmm = synth.hmm_params_i6()
    st, bad = dev.viterbi_cells(pre, cs, *hmm)
    torch.cuda.synchronize()
    assert int(bad.item()) == 0
    stats = dev.viterbi_last_stats()
    assert stats["path"] == "fast" and stats["sequences"] == 22 * C and not stats["fallback"]
    n_ref = int(sum(len(r) for r in refs))
    assert np.array_equal(np.concatenate(refs), np.arange(n_ref))      # the generator puts the reference cells first
    yield {"x": x, "cs": cs, "refs": refs, "out": out, "pre": pre, "st": st, "hmm": hmm, "n_ref": n_ref}


def oracle_slices(r):
    """For every block of observation cells: (global cell indices of the rows to compare, oracle denoised, oracle
    pre-denoise -- both as (cells, G) CUDA tensors of those rows --, (mu, s), the slice's full host pre-denoise matrix
    (G, n_ref + block), the position of the compared rows inside the slice and the block's observation cells [a, b)).  The first slice also yields the
    reference cells."""
    n_ref, x = r["n_ref"], r["x"]
    ref_rows = torch.arange(n_ref, device="cuda")
    first = True
    for a in range(n_ref, C, BLOCK):
        b = min(C, a + BLOCK)
        rows = torch.cat([ref_rows, torch.arange(a, b, device="cuda")])
        xh = x[rows].cpu().numpy().T                                   # (G, n) Fortran-ordered view, no copy
        ref_out, ref_pre, musd = oc.smooth_chain(xh, r["cs"], r["refs"], want_pre_denoise=True)
        lo = 0 if first else n_ref
        cmp_rows = rows[lo:]
        yield (cmp_rows, torch.from_numpy(ref_out.T[lo:]).cuda(), torch.from_numpy(ref_pre.T[lo:]).cuda(), musd,
               ref_pre, lo, (a, b))
        first = False


def test_config2_chain_and_states_vs_oracle_every_cell(dev, run50k):
    """(a) of the round-4 brief: |pre - ref_pre| <= 1e-11 over all 5e8 elements, every step-22 difference tied to a bound
    and counted, all 5e8 i6 state calls equal to the oracle's -- on identical inputs and end to end."""
    r = run50k
    means, sd, logPi, logDelta = r["hmm"]
    worst, flips, seen = 0.0, 0, 0
    mism_same_input, mism_end_to_end = 0, 0
    musd0 = None
    for rows, ref_out, ref_pre, musd, ref_pre_host, lo, _ in oracle_slices(r):
        musd0 = musd0 or musd
        assert musd == musd0, "the oracle's step-22 parameters must not depend on the slice"
        got_pre, got_out = r["pre"][rows], r["out"][rows]
        worst = max(worst, float((got_pre - ref_pre).abs().max()))
        flips += check_denoise_flips_t(got_out, ref_out, ref_pre, musd[0], musd[1], tol=1e-11,
                                       label=f"config 2, cells {int(rows[0])}..{int(rows[-1])}")
        got_st = r["st"][rows].cpu().numpy().T
        # identical inputs: the oracle's Viterbi on the GPU's own HMM input
        want_same, _ = oc.viterbi_cells(got_pre.cpu().numpy().T, r["cs"], means, sd, logPi, logDelta)
        mism_same_input += int((got_st != want_same).sum())
        # end to end: the oracle's Viterbi on the oracle's own chain output
        want_e2e, _ = oc.viterbi_cells(ref_pre_host[:, lo:], r["cs"], means, sd, logPi, logDelta)
        mism_end_to_end += int((got_st != want_e2e).sum())
        seen += rows.numel()
    assert seen == C
    print(f"[full size, config 2] {C} cells x {G} genes vs the oracle: chain max |delta| {worst:.2e}, {flips} denoise "
          f"selects on a bound, state mismatches {mism_same_input} (identical inputs) / {mism_end_to_end} (end to end) "
          f"of {C * G}")
    assert worst <= 1e-11
    assert flips <= 8, flips                      # 0-3 per 1e8 elements on this generator (legal ones only, see above)
    assert mism_same_input == 0
    assert mism_end_to_end == 0


def test_config4_i3_subclusters_vs_oracle_every_subcluster(dev, run50k):
    """(b): i3 HMM at subcluster level, all 102 subclusters of the 50 000 cells, against `oc.viterbi_groups` on the
    ORACLE's own chain output with the ORACLE's own i3 parameters (mu, sigma over its reference values) -- nothing of
    the GPU's arithmetic on the checker's side."""
    from infercnv_amd import synth
    r = run50k
    n_ref = r["n_ref"]
    subs, is_ref, _ = synth.subclusters(C)
    assert len(subs) == 102                                            # 10 reference + 4 x 23 clone subclusters (the last ones 250 cells)
    ref_idx = np.concatenate(r["refs"])
    mu, sigma = dev.cells_mean_sd(r["pre"], ref_idx)
    Pi, dl = onp.get_HMM_i3(1e-6)
    lPi, ldl = np.log(Pi), np.log(dl)
    delta = abs(-1.6448536269514722 * sigma)                           # |qnorm(0.05, 0, sigma)|, R/inferCNV_i3HMM.R:435-445
    st, bad = dev.viterbi_groups(r["pre"], r["cs"], subs, np.array([mu - delta, mu, mu + delta]), [sigma] * len(subs), lPi, ldl)
    torch.cuda.synchronize()
    assert int(bad.item()) == 0
    mismatches, groups_seen = 0, 0
    for _, _, _, _, ref_pre_host, lo, (a, b) in oracle_slices(r):
        o_mu, o_sigma = oc.mean_sd_of_cells(ref_pre_host, ref_idx)     # reference cells sit at 0..n_ref-1 of every slice
        assert abs(o_mu - mu) < 1e-12 and abs(o_sigma - sigma) < 1e-12
        o_delta = abs(-1.6448536269514722 * o_sigma)
        o_m3 = np.array([o_mu - o_delta, o_mu, o_mu + o_delta])
        # this slice's subclusters: the observation ones inside [a, b), plus -- first slice -- the reference ones
        mine = [q for q, g in enumerate(subs) if (a <= int(g[0]) < b) or (lo == 0 and is_ref[q])]
        local = [np.where(subs[q] < n_ref, subs[q], subs[q] - a + n_ref).astype(np.int32) for q in mine]
        assert all(int(subs[q].max()) < b for q in mine if not is_ref[q]), "a subcluster straddles two slices"
        want, _ = oc.viterbi_groups(ref_pre_host, r["cs"], local, o_m3, [o_sigma] * len(local), lPi, ldl)
        for q, loc in zip(mine, local):
            got = st[torch.as_tensor(subs[q], device="cuda")].cpu().numpy().T
            mismatches += int((got != want[:, loc]).sum())
        groups_seen += len(mine)
    assert groups_seen == len(subs)
    print(f"[full size, config 4] {len(subs)} subclusters, i3 HMM vs the oracle end to end: {mismatches} differing state calls of {C * G}")
    assert mismatches == 0


def test_config5_median_filter_vs_oracle_whole_slice(dev, run50k):
    """(c): a whole 10 000 x 5 000 slice of the denoised matrix -- two contiguous 500-cell reference tiles and eight
    interleaved (stride-4) clone subclusters -- against `oc.median_filter`, exact."""
    from infercnv_amd import synth
    r = run50k
    lo, hi = 4000, 9000                                                # 1 000 reference cells + 4 000 observation cells
    subs, _, _ = synth.subclusters(C)
    tiles = [g - lo for g in subs if lo <= int(g[0]) and int(g[-1]) < hi]
    assert len(tiles) == 10 and sum(len(t) for t in tiles) == hi - lo
    sl = r["out"][lo:hi].contiguous()
    y = dev.median_filter(sl, r["cs"], [t.astype(np.int32) for t in tiles], 7)
    want = oc.median_filter(sl.cpu().numpy().T, r["cs"], tiles, 7)
    np.testing.assert_array_equal(y.cpu().numpy().T, want)
