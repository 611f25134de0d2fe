"""The north star's own shapes (-m gpu): 10 000 genes x 250 000 and x 1 000 000 cells on ONE MI355X -- the per-device
shares of BASELINE.json configs[2] at 4 and at 1 GPU(s).  Both cross 2^31 elements (2.5e9 / 1e10): every 64-bit index,
the back-pointer batches of the Viterbi past 429 000 cells and the pool's > 16 GiB blocks run here for the first time.

The HIP path (C ABI, device-resident) is compared with the CPU ORACLE over `[all reference cells | sampled observation
cells]` exactly as tests/test_gpu_fullsize.py slices the 50 000-cell matrix: cells are independent given the reference
cells' statistics (R/inferCNV_ops.R:1678-1786, 2302-2346), so the oracle over that slice sees what the whole matrix
sees.  The sample holds the first and the LAST 100 columns, the columns on both sides of element 2^31 and of every
2^31 multiple after it, the columns on both sides of the Viterbi's batch boundary, and random ones.  The reference
share is shrunk (2 % / 1 %) so that the oracle stays well under a minute.

Tolerances: chain |delta| <= 1e-11 on the pre-denoise matrix, every step-22 difference tied to a bound
(tests/parity_util.py), i6 states bit-exact on identical inputs AND end to end, group HMM / median filter exact.
Reference semantics: R/inferCNV_ops.R:2335, R/inferCNV_HMM.R:284-324, 345-408, 1101-1176, R/noise_reduction.R:92-113.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import oracle_c as oc  # noqa: E402
from parity_util import check_denoise_flips_t  # noqa: E402

G = 10000
TWO31 = 1 << 31


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from infercnv_amd import device
    torch.cuda.set_device(0)
    device.init(0)
    device.viterbi_set_mode(0)
    return device


def _free_everything(dev):
    """Give the library's pool and torch's cache back to the driver: the 1 M-cell case needs 270 of the 288 GB."""
    import gc
    gc.collect()
    torch.cuda.synchronize()
    dev.release_pool()
    torch.cuda.empty_cache()


def _sample_columns(C, n_ref, batch_cells, n_random=1500, seed=5):
    """Observation columns to check: both ends, both sides of every multiple of 2^31 elements, both sides of the Viterbi's
    column-batch boundaries, and random ones."""
    cols = set(range(n_ref, n_ref + 100)) | set(range(C - 100, C))
    for k in range(1, (G * C) // TWO31 + 1):
        c = (k * TWO31) // G                       # the column that holds element k * 2^31
        cols.update(range(max(n_ref, c - 3), min(C, c + 4)))
    for b in range(batch_cells, C, batch_cells):
        cols.update(range(max(n_ref, b - 3), min(C, b + 4)))
    rng = np.random.default_rng(seed)
    cols.update(int(v) for v in rng.integers(n_ref, C, n_random))
    return np.array(sorted(cols), dtype=np.int64)


def _run_and_check(dev, C, ref_frac, label):
    from infercnv_amd import synth
    _free_everything(dev)
    free, total = torch.cuda.mem_get_info()
    need = 3 * 8 * G * C + G * C + (17 << 30) + (4 << 30)      # x, out, pre, states, Viterbi scratch, slack
    if need > free:                                            # a smaller scratch batch instead of failing: the paths stay the same
        budget_mb = max(1024, int((free - (3 * 8 * G * C + G * C + (4 << 30))) // (1 << 20)))
        os.environ["ICNV_VITERBI_SCRATCH_MB"] = str(budget_mb)
    batch_cells = (int(os.environ.get("ICNV_VITERBI_SCRATCH_MB", 16384)) << 20) // (G * 4) // 64 * 64
    try:
        x, cs = synth.make_matrix_torch(G, C, "cuda", ref_frac=ref_frac)
        refs, _ = synth.groups(C, ref_frac=ref_frac)
        n_ref = int(sum(len(r) for r in refs))
        assert np.array_equal(np.concatenate(refs), np.arange(n_ref))
        out, pre = dev.smooth_chain(x, cs, refs, want_pre_denoise=True)
        hmm = synth.hmm_params_i6()
        st, bad = dev.viterbi_cells(pre, cs, *hmm)
        torch.cuda.synchronize()
        assert int(bad.item()) == 0
        stats = dev.viterbi_last_stats()
        assert stats["path"] == "fast" and not stats["fallback"], stats
        assert stats["sequences"] == 22 * C

        cols = _sample_columns(C, n_ref, batch_cells)
        rows = torch.cat([torch.arange(n_ref, device="cuda"), torch.as_tensor(cols, device="cuda")])
        xh = x[rows].cpu().numpy().T                                       # (G, n_ref + sample), column-major view
        ref_out, ref_pre, (mu, s) = oc.smooth_chain(xh, cs, refs, want_pre_denoise=True)
        got_pre, got_out = pre[rows], out[rows]
        r_pre = torch.from_numpy(ref_pre.T).cuda()
        worst = float((got_pre - r_pre).abs().max())
        flips = check_denoise_flips_t(got_out, torch.from_numpy(ref_out.T).cuda(), r_pre, mu, s, tol=1e-11, label=label)
        means, sd, logPi, logDelta = hmm
        got_st = st[rows].cpu().numpy().T
        want_same, _ = oc.viterbi_cells(got_pre.cpu().numpy().T, cs, means, sd, logPi, logDelta)
        want_e2e, _ = oc.viterbi_cells(ref_pre, cs, means, sd, logPi, logDelta)
        m_same, m_e2e = int((got_st != want_same).sum()), int((got_st != want_e2e).sum())
        print(f"[{label}] {C} cells x {G} genes ({G * C:.3e} elements, {G * C // TWO31} multiples of 2^31 crossed), oracle over "
              f"{n_ref} reference + {cols.size} sampled observation cells: chain max |delta| {worst:.2e}, {flips} denoise selects "
              f"on a bound, state mismatches {m_same} (identical inputs) / {m_e2e} (end to end) of {got_st.size}; "
              f"Viterbi batches of {batch_cells} cells, {stats['flagged']} sequences redone exactly in the last batch")
        assert worst <= 1e-11
        assert flips <= 8
        assert m_same == 0 and m_e2e == 0
        # every state byte of the 1e9..1e10 was written: states are 1..6, nothing of the buffer is left untouched
        lo_hi = torch.stack([st.min(), st.max()]).cpu().tolist()
        assert 1 <= lo_hi[0] and lo_hi[1] <= 6, lo_hi
        return {"x": x, "cs": cs, "refs": refs, "n_ref": n_ref, "out": out, "pre": pre, "st": st, "hmm": hmm}
    finally:
        os.environ.pop("ICNV_VITERBI_SCRATCH_MB", None)


def test_250k_cells_chain_viterbi_groups_median_past_2_31_elements(dev):
    """10 000 x 250 000 (config 3's share of one of FOUR GPUs): chain + per-cell i6 Viterbi against the oracle, then the
    group HMM and the median filter on subclusters / tiles that straddle element 2^31 (column 214 748) and on the last ones."""
    from infercnv_amd import synth
    C = 250000
    r = _run_and_check(dev, C, 0.02, "250k cells")
    n_ref = r["n_ref"]
    subs, is_ref, _ = synth.subclusters(C, ref_frac=0.02)
    edge = TWO31 // G                                                    # 214 748: holds element 2^31
    # observation subclusters: a block of all four clones spans 2 000 consecutive columns; the block around `edge` and the last one
    picked = [q for q, g in enumerate(subs) if not is_ref[q] and ((int(g[0]) <= edge <= int(g[-1])) or int(g[-1]) >= C - 4)]
    assert len(picked) >= 5, picked
    lo = min(int(subs[q][0]) for q in picked if int(subs[q][0]) <= edge)
    span = np.arange(lo, min(C, lo + 2200))                               # the straddling block's columns ...
    last = np.arange(min(int(subs[q][0]) for q in picked if int(subs[q][-1]) >= C - 4), C)
    keep = np.unique(np.concatenate([span, last]))
    groups = [subs[q] for q in picked if np.isin(subs[q], keep).all()]
    assert groups and any(int(g[0]) <= edge <= int(g[-1]) for g in groups)
    # ---- group HMM (i6 parameters, one sd per group) on the FULL matrix, oracle on the compacted columns
    means, sd, logPi, logDelta = r["hmm"]
    st_g, bad = dev.viterbi_groups(r["pre"], r["cs"], groups, means, [sd] * len(groups), logPi, logDelta)
    torch.cuda.synchronize()
    assert int(bad.item()) == 0
    pos = {int(c): i for i, c in enumerate(keep)}
    local = [np.array([pos[int(c)] for c in g], dtype=np.int32) for g in groups]
    sub_pre = r["pre"][torch.as_tensor(keep, device="cuda")].cpu().numpy().T
    want, _ = oc.viterbi_groups(sub_pre, r["cs"], local, means, [sd] * len(groups), logPi, logDelta)
    mism = 0
    for g, loc in zip(groups, local):
        got = st_g[torch.as_tensor(g, device="cuda")].cpu().numpy().T
        mism += int((got != want[:, loc]).sum())
    print(f"[250k cells] group HMM on {len(groups)} subclusters around element 2^31 and at the end: {mism} differing state calls")
    assert mism == 0
    del st_g
    # ---- median filter: the same subclusters as tiles (+ nothing else: the other cells pass through)
    y = dev.median_filter(r["out"], r["cs"], [g.astype(np.int32) for g in groups], 7)
    torch.cuda.synchronize()
    sub_out = r["out"][torch.as_tensor(keep, device="cuda")].cpu().numpy().T
    want_mf = oc.median_filter(sub_out, r["cs"], local, 7)
    got_mf = y[torch.as_tensor(keep, device="cuda")].cpu().numpy().T
    in_tiles = np.unique(np.concatenate(local))
    np.testing.assert_array_equal(got_mf[:, in_tiles], want_mf[:, in_tiles])
    # cells in no tile pass through unchanged -- checked on the last reference column and on a column before the straddling block
    for c in (n_ref - 1, lo - 1):
        assert torch.equal(y[c], r["out"][c])
    del y, r
    _free_everything(dev)


def test_1m_cells_chain_and_viterbi_on_one_gpu(dev):
    """10 000 x 1 000 000: BASELINE.json configs[2] on ONE GPU (80 GB in, 80 + 80 out, 10 GB of states, 16 GB of
    back-pointer scratch per 429 000-cell batch: three Viterbi batches)."""
    r = _run_and_check(dev, 1000000, 0.01, "1M cells")
    del r
    _free_everything(dev)
