#!/usr/bin/env python3
"""Soak run of the host-buffer boundary (run on the GPU box): random shapes through the step functions an R session would
call -- fused chain, per-cell i6 HMM, subcluster HMM, median filter -- every result checked against the CPU oracle, with
argument errors thrown in between (the library must report them and keep working) and the device memory the library
holds tracked from call to call: the same sequence of shapes runs twice, and the second pass must not need more device
memory than the first (a pool that grows with the number of calls, not with the shapes, fails here).
  python tests/campaigns/soak.py [n_iterations_per_pass] [seed]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root, os.path.join(root, "tests"), os.path.join(root, "oracle")]
import numpy as np, torch
import oracle_c as oc
import oracle_np as onp
from parity_util import check_denoise_flips
from infercnv_amd import GeneOrder, InfercnvObject, device, hmm, noise_reduction, ops, _lib

torch.cuda.set_device(0); device.init(0)
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 150
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def used_mb():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2 ** 20


CNV = {k: {"mean": m, "sd": s} for k, m, s in zip(hmm.CNV_LEVELS, (0.41, 0.84, 1.017, 1.122, 1.238, 1.443), (0.03, 0.16, 0.11, 0.19, 0.24, 0.29))}
MEANS = [CNV[k]["mean"] for k in hmm.CNV_LEVELS]
SD = float(onp.r_median(np.array([CNV[k]["sd"] for k in hmm.CNV_LEVELS])))
Pi, delta = onp.get_HMM_i6(1e-6)
t0 = time.time()
base = used_mb()
traces, errors_seen, values_checked = [[], []], 0, 0
for it in range(2 * n_it):
    if it % n_it == 0: rng = np.random.default_rng(seed0)    # second pass: the same shapes and data again
    trace = traces[it // n_it]
    n_chr = int(rng.integers(1, 24))
    sizes = rng.integers(1, int(rng.choice([30, 200, 900])), size=n_chr)
    G = int(sizes.sum())
    C = int(rng.integers(8, int(rng.choice([60, 400, 2500]))))
    n_ref = int(rng.integers(2, max(3, C // 3)))
    perm = rng.permutation(C)
    ref_all = perm[:n_ref]
    cut = int(rng.integers(1, n_ref)) if rng.random() < 0.6 else n_ref
    refs = {"r0": np.sort(ref_all[:cut]).astype(np.int32)}
    if cut < n_ref: refs["r1"] = ref_all[cut:].astype(np.int32)          # (unsorted, as R allows)
    obs_all = perm[n_ref:]
    obs = {"t0": obs_all.astype(np.int32)}
    chr_codes = np.repeat(np.arange(n_chr), sizes)
    x = rng.normal(0.0, 0.6, size=(G, C)) + rng.normal(2.0, 0.5, size=(G, 1))
    if obs_all.size > 4:
        g0 = int(rng.integers(0, G)); x[g0:g0 + G // 3, obs_all[: obs_all.size // 2]] += 0.8
    x = np.ascontiguousarray(np.abs(x))
    obj = InfercnvObject(expr_data=x, gene_order=GeneOrder(chr=np.array(["chr%d" % (c + 1) for c in chr_codes])),
                         reference_grouped_cell_indices=refs, observation_grouped_cell_indices=obs)
    cs = oc.chr_starts_from_codes(chr_codes)
    W = int(rng.choice([101, 101, 51, 11, 3]))
    ref_list = list(refs.values())
    want_out, want_pre, musd = oc.smooth_chain(x, cs, ref_list, window_length=W, want_pre_denoise=True)
    fused, hmm_in = ops.hip_smooth_chain(obj, window_length=W, return_hmm_input=True)
    assert np.abs(hmm_in.expr_data - want_pre).max() < 1e-11, (it, "chain")
    check_denoise_flips(fused.expr_data, want_out, want_pre, *musd, tol=1e-11, label="soak %d" % it)
    cells = hmm.predict_CNV_via_HMM_on_indiv_cells(hmm_in, CNV)
    want_states, bad = oc.viterbi_cells(hmm_in.expr_data, cs, MEANS, SD, np.log(Pi), np.log(delta))
    assert bad == 0 and np.array_equal(cells.expr_data, want_states), (it, "viterbi")
    # subclusters of the observation group + the reference groups as tiles of the median filter
    k = int(rng.integers(1, 5))
    parts = [p.astype(np.int32) for p in np.array_split(obs_all, k) if p.size]
    fused.tumor_subclusters = {"subclusters": {"t0": {"s%d" % i: p for i, p in enumerate(parts)},
                                               **{n: {n: v} for n, v in refs.items()}}}
    mf = noise_reduction.apply_median_filtering(fused)
    tiles = parts + ref_list
    want_mf = oc.median_filter(fused.expr_data, cs, tiles, 7)
    assert np.array_equal(mf.expr_data, want_mf), (it, "median filter")
    values_checked += 4 * G * C
    if it % 5 == 2:
        # argument errors: the library reports them (exception with its message) and the next call works
        for breaker in (lambda: ops.hip_smooth_chain(obj, window_length=4),                       # even window
                        lambda: hmm._viterbi_cells(hmm_in.expr_data, cs[::-1].copy(), MEANS, SD, Pi, delta),   # decreasing offsets
                        lambda: device.median_filter(torch.zeros((4, 4), dtype=torch.float64, device="cuda"), np.array([0, 4], np.int32),
                                                     [np.array([0, 9], np.int32)], 7)):          # cell index out of range
            try:
                breaker()
            except Exception:   # noqa: BLE001 -- any error type the binding raises
                errors_seen += 1
            else:
                raise AssertionError("iteration %d: an invalid call was accepted" % it)
    trace.append(used_mb())

device.release_pool()
after = used_mb()
torch.cuda.empty_cache()
after_torch = used_mb()
print("soak: 2 x %d iterations (seed %d), %d values identical / within 1e-11 of the oracle, %d injected argument errors reported, %.0f s"
      % (n_it, seed0, values_checked, errors_seen, time.time() - t0))
print("device memory in use (MiB): before %.0f; first pass max %.0f, end %.0f; second pass (same shapes) max %.0f, end %.0f; "
      "after release_pool() %.0f, after torch.cuda.empty_cache() as well %.0f"
      % (base, max(traces[0]), traces[0][-1], max(traces[1]), traces[1][-1], after, after_torch))
assert max(traces[1]) <= max(traces[0]) + 64, "the second pass over the same shapes needed more device memory than the first"
