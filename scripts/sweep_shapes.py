#!/usr/bin/env python3
"""Where the two hot kernels stand away from the bench's one (G, sigma) point (run on the GPU box):
   * the fused smooth pass and the per-cell i6 Viterbi at G in {6k .. 20k} genes (22 chromosomes of the bench's
     proportions, 20 000 cells): ms, cells/s, GB/s on the algorithmic bytes, the chain geometry that served it;
   * the Viterbi at 10 000 genes for shared sds from 0.05 to 0.3 (the i6 means of data/mcmc_obj.rda): path taken
     (certified fast kernel / exact kernel), flagged sequences, table records.
   python scripts/sweep_shapes.py > profiles/r05_sweep.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from infercnv_amd import device, synth

torch.cuda.set_device(0); device.init(0)
C = 20000
res = {"cells": C, "genes": [], "sigma": []}


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def geometry(G, n_chr=22, T=50):
    pad = (T + 3) & ~1
    npos = G + (n_chr + 1) * pad
    if npos <= 768 * 7: return "768x7", npos
    if npos <= 1024 * 11 and G <= 10240: return "1024x11", npos
    if npos <= 768 * 15: return "768x15", npos
    for L in (17, 19, 21):
        if npos <= 768 * L and G <= 768 * ((L - 1) // 2) * 2: return f"768x{L}", npos
    if npos <= 768 * 23: return "768x23", npos
    if npos <= 512 * 35: return "512x35", npos
    return ("three-pass (2T+1 taps)" if os.environ.get("ICNV_CHAIN_LARGE_TAPS") == "1" else "two-pass (views + centre/finish)"), npos


means, sd, logPi, logDelta = synth.hmm_params_i6()
for G in [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else "6000,8000,9939,10000,11000,12000,14000,16000,18000,20000,24000".split(","))]:
    x, cs = synth.make_matrix_torch(G, C, "cuda")
    refs, _ = synth.groups(C)
    out, pre = torch.empty_like(x), torch.empty_like(x)
    plan = device.ChainPlan(G, C, cs, refs)
    def chain_step():
        for r in range(plan.num_rounds):
            plan.round_partial(r, x); plan.round_finish(r)
        plan.apply(x, out=out, pre=pre)
    t_chain = timed(chain_step)
    device.timing_reset(); device.timing_enable(True)
    chain_step(); torch.cuda.synchronize()
    k_apply = device.timing_get("chain_apply")[0] + device.timing_get("chain_apply_ref")[0]
    device.timing_enable(False)
    states = torch.empty((C, G), dtype=torch.uint8, device="cuda")
    t_vit = timed(lambda: device.viterbi_cells(pre, cs, means, sd, logPi, logDelta, states=states))
    st = device.viterbi_last_stats()
    geo, npos = geometry(G)
    res["genes"].append({"G": G, "padded_positions": npos, "chain_geometry": geo,
                         "chain_ms_incl_reference_rounds": t_chain * 1e3, "chain_apply_kernels_ms": k_apply,
                         "chain_cells_per_s": C / t_chain, "chain_GBps_algorithmic_16B": 16.0 * G * C / t_chain / 1e9,
                         "viterbi_ms": t_vit * 1e3, "viterbi_cells_per_s": C / t_vit, "viterbi_GBps_algorithmic_9B": 9.0 * G * C / t_vit / 1e9,
                         "viterbi_path": st["path"], "viterbi_flagged": st["flagged"],
                         "step_gene_cells_per_s": G * C / (t_chain + t_vit)})
    if G % 16:
        # the same step with the HMM input and the states in a padded layout (leading dimension = G rounded up to 16: every cell on
        # a cache line / a 16-byte word of its own; icnv_chain_apply_ld_dev + icnv_viterbi_cells_ld_dev)
        pre_p = device.padded_matrix(C, G)
        st_p = device.padded_matrix(C, G, torch.uint8)
        def chain_step_p():
            for r in range(plan.num_rounds):
                plan.round_partial(r, x); plan.round_finish(r)
            plan.apply(x, out=out, pre=pre_p)
        t_chain_p = timed(chain_step_p)
        t_vit_p = timed(lambda: device.viterbi_cells(pre_p, cs, means, sd, logPi, logDelta, states=st_p))
        assert torch.equal(st_p, states) and torch.equal(pre_p, pre)
        res["genes"][-1].update({"padded_ld": int(pre_p.stride(0)), "chain_ms_padded_hmm_input": t_chain_p * 1e3, "viterbi_ms_padded": t_vit_p * 1e3,
                                 "step_gene_cells_per_s_padded": G * C / (t_chain_p + t_vit_p)})
        del pre_p, st_p
    plan.close()
    del x, out, pre, states, plan

G = 10000
x, cs = synth.make_matrix_torch(G, C, "cuda")
refs, _ = synth.groups(C)
_, pre = device.smooth_chain(x, cs, refs, want_pre_denoise=True)
states = torch.empty((C, G), dtype=torch.uint8, device="cuda")
for s in (0.03, 0.05, 0.07, 0.1, 0.18, 0.3, 0.5):
    t = timed(lambda: device.viterbi_cells(pre, cs, means, s, logPi, logDelta, states=states), 3)
    st = device.viterbi_last_stats()
    res["sigma"].append({"sd": s, "viterbi_ms": t * 1e3, "cells_per_s": C / t, "path": st["path"], "fallback_to_exact": st["fallback"],
                         "flagged_sequences": st["flagged"], "sequences": st["sequences"], "table_records": st["table_intervals"]})
print(json.dumps(res))
