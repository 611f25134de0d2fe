#!/usr/bin/env python3
"""Benchmark of the hot path named by BASELINE.json: cells/sec through the fused
smoothing chain (steps 8,9,10,11,12,14,22) + per-cell i6 HMM Viterbi at
10 000 genes, cell-sharded over N MI355X (one process per GPU, RCCL).

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over this rank's resident batch of cells:
3 reference rounds (partial -> all-reduce -> finish), the fused apply pass
writing the denoised matrix + the HMM input, and the Viterbi writing uint8
states.  Inputs are synthetic (infercnv_amd/synth.py), generated in HBM before
the timed region.  Weak scaling: every rank holds --cells cells (default
50 000 = BASELINE.json configs[1]).

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the field meanings).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured copy ceiling
def _stream_ceiling_from_file():
    """Fallback only (the ubench binaries are missing): the best "1r2w" line of profiles/ubench_stream_1r2w.txt, the tracked
    output of scripts/ubench/stream_1r2w.hip on SOME MI355X on some other day; 5300 GB/s if the file is absent too."""
    best = None
    try:
        for line in open(os.path.join(ROOT, "profiles", "ubench_stream_1r2w.txt")):
            if line.startswith("1r2w") and "TB/s" in line:
                v = float(line.split()[-2]) * 1000.0
                best = v if best is None else max(best, v)
    except Exception:
        pass
    return (best, "profiles/ubench_stream_1r2w.txt (NOT measured in this run)") if best else (5300.0, "constant (no ubench binary, no profile)")


def measure_ceilings():
    """The two memory-system ceilings the roofline objects are priced against, measured ON THIS BOX IN THIS RUN, before any
    bench tensor exists (boxes differ by +-10 %): scripts/ubench/stream_1r2w (a plain grid-stride 1-read : 2-write stream --
    the traffic mix of the fused smooth pass) and scripts/ubench/column_walk (every lane walks a column of its own, 128-byte
    visits, 768 lanes per CU -- the Viterbi's observation stream without arithmetic), both in their `quick` mode (< 1 s).
    Built by __graft_entry__.build() (hipcc, in-tree, git-ignored).  Returns a dict; falls back to the tracked files."""
    import subprocess
    res = {}
    cache = {}
    def run(name):
        if name in cache: return cache[name]
        cache[name] = _run(name)
        return cache[name]
    def _run(name):
        exe = os.path.join(ROOT, "scripts", "ubench", name)
        out = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=120, cwd="/tmp")
        if out.returncode != 0:
            raise RuntimeError(out.stderr[-200:])
        return out.stdout
    try:
        best = None
        for line in run("stream_1r2w").splitlines():
            if line.startswith("1r2w") and "TB/s" in line:
                v = float(line.split()[-2]) * 1000.0
                best = v if best is None else max(best, v)
        if not best:
            raise RuntimeError("no 1r2w line")
        res["stream_1r2w_gbs"], res["stream_1r2w_source"] = best, "scripts/ubench/stream_1r2w quick, measured in this run on this GPU"
    except Exception as e:
        v, src = _stream_ceiling_from_file()
        res["stream_1r2w_gbs"], res["stream_1r2w_source"] = v, f"{src}; in-run measurement failed: {str(e)[:80]}"
    try:
        best = None
        for line in run("column_walk").splitlines():
            if line.startswith("visit") and "TB/s" in line:
                v = float(line.split()[-2]) * 1000.0
                best = v if best is None else max(best, v)
        if not best:
            raise RuntimeError("no visit line")
        res["column_walk_gbs"], res["column_walk_source"] = best, "scripts/ubench/column_walk quick (768 lanes per CU, 128-byte visits), measured in this run on this GPU"
    except Exception as e:
        res["column_walk_gbs"] = 3660.0
        res["column_walk_source"] = f"profiles/r04_ubench_column_walk.txt (NOT measured in this run: {str(e)[:80]})"
    try:
        # the same lines requested by rows through LDS-DMA: what the staged fast Viterbi's observation stream can reach
        best = None
        for line in run("column_walk").splitlines():
            if line.startswith("rows via LDS-DMA") and "TB/s" in line:
                v = float(line.split()[-2]) * 1000.0
                best = v if best is None else max(best, v)
        if best:
            res["row_requests_gbs"], res["row_requests_source"] = best, "scripts/ubench/column_walk quick (whole 128-byte lines by rows through LDS-DMA, 768 lanes per CU), measured in this run on this GPU"
    except Exception:
        pass
    return res


STREAM_1R2W_GBS, STREAM_1R2W_SOURCE = _stream_ceiling_from_file()     # replaced by measure_ceilings() in main()
FP64_VECTOR_PEAK_TF = 78.6   # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
VITERBI_VALU_PER_GENE = 92.1   # SQ_INSTS_VALU / (genes x cells / 64) of the staged kernel, profiles/r06_pmc_viterbi_fast.txt: 7.1976e8 / 7.8125e6 (traceback included; round 5 measured the same)
COLUMN_WALK_GBS, COLUMN_WALK_SOURCE = 3660.0, "profiles/r04_ubench_column_walk.txt"   # replaced by measure_ceilings() in main()
ROW_REQUESTS_GBS, ROW_REQUESTS_SOURCE = None, None                                     # set by measure_ceilings() in main()


def source_stamp():
    """16 hex digits of the sha256 over the library's sources (infercnv_amd/csrc, include/): what a counter file under
    profiles/ is tied to.  scripts/pmc_summary.py writes it (and the commit, when it is told one) into
    profiles/pmc_traffic.json; a bench line only quotes counter traffic collected on THIS source."""
    import hashlib
    h = hashlib.sha256()
    files = []
    for d in (os.path.join(ROOT, "infercnv_amd", "csrc"), os.path.join(ROOT, "include")):
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".inc", ".h", ".cpp")) or f == "Makefile":
                files.append(os.path.join(d, f))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _pmc_traffic():
    """profiles/pmc_traffic.json (scripts/pmc_traffic.sh + pmc_summary.py): HBM bytes from separate rocprofv3 --pmc passes.
    Returns ({} if absent) the file's content plus "_current": whether its source stamp is this tree's."""
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        return {}
    tr["_current"] = tr.get("source_sha16") == source_stamp()
    return tr


def _traffic_fields(tr):
    """traffic_source / traffic_commit / traffic_source_sha16 of a roofline object."""
    return {"traffic_source": "profiles/pmc_traffic.json: FETCH_SIZE (x2 as MI355X_MICROARCH.md prescribes; calibrated on the chain's "
                              "16-byte coalesced stream, uncalibrated for the Viterbi's per-lane lines) + WRITE_SIZE of separate "
                              "rocprofv3 --pmc passes over this workload (scripts/pmc_traffic.sh), per launch; not measured in this run"
                              + ("" if tr.get("_current") else " -- STALE: collected on other sources than this tree's, traffic set to null"),
            "traffic_commit": tr.get("commit"), "traffic_source_sha16": tr.get("source_sha16"),
            "traffic_is_of_this_source": bool(tr.get("_current"))}


def effective_cores():
    """CPU cores this process may actually use: the affinity mask capped by the container's CPU quota (cgroup v2
    cpu.max, cgroup v1 cfs quota).  os.cpu_count() reports the node's logical CPUs, whatever the quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    eff = min(float(n), quota) if quota else float(n)
    return max(1.0, eff), n, quota


def cpu_baseline(G, target_seconds=12.0, on_gpu=True):
    """The plain-C oracle (oracle/icnv_oracle.c, OpenMP over cells) timed on this
    host's cores over a bounded sample of the same synthetic workload.  R is not
    installed in the image, so the reference itself cannot be timed: kind="port".

    Returns (cpu_baseline, parity).  The oracle's outputs for the sample are not thrown away: the HIP path runs over the
    very same sample matrix (outside every timed region) and `parity` reports the comparison -- chain max |delta| on the
    pre-denoise matrix, the step-22 selects that fall on a bound (each one checked to be a legal flip of the strict
    select, R/inferCNV_ops.R:2335), and the number of differing i6 state calls, end to end (GPU chain -> GPU Viterbi
    against oracle chain -> oracle Viterbi)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle_c as oc
    from infercnv_amd import synth
    oc.build()
    eff, affinity, quota = effective_cores()
    cores = max(1, int(round(eff)))            # OpenMP threads = the cores the quota lets us keep busy
    oc.set_num_threads(cores)
    hmm = synth.hmm_params_i6()
    means, sd, logPi, logDelta = hmm
    if on_gpu:
        import torch
        from infercnv_amd import device

    def sample(C):
        """(host matrix (G, C) column-major, chr_start, device copy or None): generated in HBM and downloaded when a GPU
        is there (the same splitmix64 / Box-Muller generator as the bench matrix; seconds instead of a minute)."""
        if on_gpu:
            xd, cs = synth.make_matrix_torch(G, C, "cuda")
            return xd.cpu().numpy().T, cs, xd
        x, cs = synth.make_matrix_np(G, C)
        return x, cs, None

    def run(C, keep=False):
        x, cs, xd = sample(C)
        refs, _ = synth.groups(C)
        t0 = time.perf_counter()
        out, pre, musd = oc.smooth_chain(x, cs, refs, want_pre_denoise=True)
        st, _ = oc.viterbi_cells(pre, cs, means, sd, logPi, logDelta)
        t = time.perf_counter() - t0
        return (t, (xd, cs, refs, out, pre, musd, st)) if keep else (t, None)

    # pilot sized so that every thread gets several cells, then scale to ~target_seconds of CPU work
    pilot_c = max(512, 8 * cores)
    run(pilot_c)                                   # first call also pays thread start-up / page faults
    t_pilot, _ = run(pilot_c)
    C = int(min(60000, max(pilot_c, pilot_c * target_seconds / max(t_pilot, 1e-3))))
    t, kept = run(C, keep=True)
    if t < 0.6 * target_seconds and C < 60000:     # the pilot under-estimated the parallel speed: one larger sample
        C = int(min(60000, C * target_seconds / max(t, 1e-3)))
        kept = None
        t, kept = run(C, keep=True)
    res = {"value": C / t, "unit": "cells/s", "cores": oc.num_threads(), "cores_note": (
               f"{oc.num_threads()} OpenMP threads = the CPU quota of this container ({quota:.1f} cores) "
               f"of {os.cpu_count()} logical CPUs on the node" if quota else
               f"{oc.num_threads()} OpenMP threads = affinity mask ({affinity}) of {os.cpu_count()} logical CPUs, no CPU quota"),
           "kind": "port",
           "sample": f"{G} genes x {C} cells of the same synthetic generator, smooth chain + i6 Viterbi, "
                     f"oracle/icnv_oracle.c with OpenMP over cells, {t:.1f} s"}
    parity = None
    if on_gpu:
        xd, cs, refs, ref_out, ref_pre, (mu, s), ref_st = kept
        out, pre = device.smooth_chain(xd, cs, refs, want_pre_denoise=True)
        st, bad = device.viterbi_cells(pre, cs, means, sd, logPi, logDelta)
        torch.cuda.synchronize()
        r_pre = torch.from_numpy(ref_pre.T).cuda()             # (C, G) views of the oracle's column-major matrices
        r_out = torch.from_numpy(ref_out.T).cuda()
        tol = 1e-11 * max(1.0, float(r_pre.abs().max()))
        chain_max_abs = float((pre - r_pre).abs().max())
        diff = (out - r_out).abs() > tol
        flips = int(diff.sum())
        legal = True
        if flips:                                              # every differing select sits on a bound and holds a legal value
            p, g = r_pre[diff], out[diff]
            on_bound = torch.minimum((p - (mu - s)).abs(), (p - (mu + s)).abs()) <= 2.0 * tol
            legal = bool((on_bound & (((g - mu).abs() <= tol) | ((g - p).abs() <= tol))).all())
        r_st = torch.from_numpy(ref_st.T).cuda()
        bad_cells = torch.nonzero((st != r_st).any(dim=1)).flatten()
        mism = int((st != r_st).sum())
        same_input = 0
        if mism:                                               # the contract is bit-exactness on IDENTICAL inputs: redo those cells
            want, _ = oc.viterbi_cells(pre[bad_cells].cpu().numpy().T, cs, means, sd, logPi, logDelta)
            same_input = int((st[bad_cells].cpu().numpy().T != want).sum())
        parity = {"cells": C, "genes": G, "chain_max_abs": chain_max_abs, "chain_tolerance_abs": 1e-11,
                  "chain_max_rel": chain_max_abs / max(float(r_pre.abs().max()), 1e-300), "north_star_rel_tolerance": 1e-5,
                  "denoise_flips": flips, "denoise_flips_all_on_a_bound_and_legal": legal,
                  "state_calls": C * G, "state_mismatches": mism, "state_mismatches_on_identical_inputs": same_input,
                  "viterbi_sequences_redone_exactly": int(device.viterbi_last_stats()["flagged"]),
                  "ok": bool(chain_max_abs <= 1e-11 and legal and same_input == 0 and int(bad.item()) == 0),
                  "what": "HIP path (C ABI, device-resident) vs the oracle's outputs for the cpu_baseline sample: the same matrix, "
                          "every cell; outside the timed region"}
        del kept, xd, out, pre, st, r_pre, r_out, r_st
    # the reference's own structure is serial R: one core of the same port, a few seconds
    oc.set_num_threads(1)
    c1 = 512
    t1, _ = run(c1)
    oc.set_num_threads(cores)
    res["single_thread"] = {"value": c1 / t1, "unit": "cells/s", "cores": 1,
                            "sample": f"{G} genes x {c1} cells, same code on one core, {t1:.1f} s"}
    res["parallel_speedup_over_one_core"] = res["value"] / res["single_thread"]["value"]
    return res, parity


def cpu_baseline_on_matrix(x, out, pre, states, chr_start, refs, hmm, n_sample=2000, seed=11):
    """Config 3 (one GPU holds the whole job): the oracle over `[all reference cells | n_sample observation cells]` of the
    bench matrix itself -- the first and the last 100 columns, both sides of every multiple of 2^31 elements, random ones.
    Cells are independent given the reference cells' statistics (R/inferCNV_ops.R:1678-1786, 2302-2346), so the slice sees
    what the whole matrix sees.  Timed, it is the cpu_baseline sample; compared with the last step's outputs (denoised
    matrix, HMM input, states), it is the parity block.  Returns (cpu_baseline, parity)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import torch
    import oracle_c as oc
    from infercnv_amd import device
    oc.build()
    eff, affinity, quota = effective_cores()
    cores = max(1, int(round(eff)))
    oc.set_num_threads(cores)
    means, sd, logPi, logDelta = hmm
    C, G = x.shape
    ref_all = np.concatenate(refs)
    is_ref = np.zeros(C, dtype=bool)
    is_ref[ref_all] = True
    obs = np.flatnonzero(~is_ref)
    pick = set(obs[:100].tolist()) | set(obs[-100:].tolist())
    for k in range(1, (G * C) // (1 << 31) + 1):
        c = (k << 31) // G
        pick.update(int(v) for v in range(max(0, c - 2), min(C, c + 3)) if not is_ref[v])
    rng = np.random.default_rng(seed)
    pick.update(int(v) for v in rng.choice(obs, size=min(n_sample, obs.size), replace=False))
    cols = np.array(sorted(pick), dtype=np.int64)
    rows = torch.as_tensor(np.concatenate([ref_all.astype(np.int64), cols]), device="cuda")
    # the slice's reference groups: positions 0 .. n_ref-1 in the order of the concatenated groups
    off = np.concatenate([[0], np.cumsum([len(r) for r in refs])])
    refs_slice = [np.arange(off[i], off[i + 1], dtype=np.int32) for i in range(len(refs))]
    n = int(rows.numel())
    CH = 8192                                            # rows per transfer / comparison: the GPU holds 270 GB of bench tensors
    x_rows = np.empty((n, G), dtype=np.float64)          # (cells, genes) row-major == genes x cells column-major
    for a in range(0, n, CH):
        x_rows[a:a + CH] = x[rows[a:a + CH]].cpu().numpy()
    xh = x_rows.T
    t0 = time.perf_counter()
    ref_out, ref_pre, (mu, s) = oc.smooth_chain(xh, chr_start, refs_slice, want_pre_denoise=True)
    ref_st, _ = oc.viterbi_cells(ref_pre, chr_start, means, sd, logPi, logDelta)
    t = time.perf_counter() - t0
    del x_rows, xh
    base = {"value": n / t, "unit": "cells/s", "cores": oc.num_threads(), "cores_note": (
                f"{oc.num_threads()} OpenMP threads = the CPU quota of this container ({quota:.1f} cores) of {os.cpu_count()} logical CPUs on the node"
                if quota else f"{oc.num_threads()} OpenMP threads = affinity mask ({affinity}) of {os.cpu_count()} logical CPUs, no CPU quota"),
            "kind": "port",
            "sample": f"{G} genes x {n} cells of the bench matrix itself ({ref_all.size} reference + {cols.size} sampled observation cells of "
                      f"{C}), smooth chain + i6 Viterbi, oracle/icnv_oracle.c with OpenMP over cells, {t:.1f} s"}
    device.release_pool()                                # the Viterbi's 16 GiB of scratch: the comparison needs a little room
    scale = max(1.0, float(np.abs(ref_pre).max()))
    tol = 1e-11 * scale
    chain_max_abs, flips, legal, mism, same_input = 0.0, 0, True, 0, 0
    ref_pre_t, ref_out_t, ref_st_t = ref_pre.T, ref_out.T, ref_st.T      # (cells, genes) views
    for a in range(0, n, CH):
        rr = rows[a:a + CH]
        g_pre, g_out, g_st = pre[rr], out[rr], states[rr]
        r_pre = torch.from_numpy(np.ascontiguousarray(ref_pre_t[a:a + CH])).cuda()
        r_out = torch.from_numpy(np.ascontiguousarray(ref_out_t[a:a + CH])).cuda()
        chain_max_abs = max(chain_max_abs, float((g_pre - r_pre).abs().max()))
        diff = (g_out - r_out).abs() > tol
        nf = int(diff.sum())
        flips += nf
        if nf:
            pv, gv = r_pre[diff], g_out[diff]
            on_bound = torch.minimum((pv - (mu - s)).abs(), (pv - (mu + s)).abs()) <= 2.0 * tol
            legal = legal and bool((on_bound & (((gv - mu).abs() <= tol) | ((gv - pv).abs() <= tol))).all())
        r_st = torch.from_numpy(np.ascontiguousarray(ref_st_t[a:a + CH])).cuda()
        bad_rows = (g_st != r_st).any(dim=1)
        nm = int((g_st != r_st).sum())
        mism += nm
        if nm:                                           # the contract is bit-exactness on IDENTICAL inputs: redo those cells
            bad_cells = torch.nonzero(bad_rows).flatten()
            want, _ = oc.viterbi_cells(g_pre[bad_cells].cpu().numpy().T, chr_start, means, sd, logPi, logDelta)
            same_input += int((g_st[bad_cells].cpu().numpy().T != want).sum())
        del g_pre, g_out, g_st, r_pre, r_out, r_st, diff
    parity = {"cells": int(n), "of_cells": int(C), "genes": int(G), "elements_of_the_matrix": int(G) * int(C),
              "multiples_of_2_31_elements_crossed": int((G * C) >> 31),
              "chain_max_abs": chain_max_abs, "chain_tolerance_abs": 1e-11,
              "chain_max_rel": chain_max_abs / scale, "north_star_rel_tolerance": 1e-5,
              "denoise_flips": flips, "denoise_flips_all_on_a_bound_and_legal": legal,
              "state_calls": int(n) * int(G), "state_mismatches": mism, "state_mismatches_on_identical_inputs": same_input,
              "ok": bool(chain_max_abs <= 1e-11 and legal and same_input == 0),
              "what": "the timed step's own outputs (C ABI, device-resident) vs the oracle over [all reference cells | sampled observation "
                      "cells incl. both ends and both sides of every 2^31-element multiple] of the same matrix; outside the timed region"}
    return base, parity


def host_path_rate(x_dev, chr_start, refs, hmm, cells=20000):
    """PCIe-inclusive rate of the HOST-BUFFER entry points -- what an R process gets through the .Call shim: the fused
    icnv_smooth_chain (matrix up; denoised matrix + HMM input down) and icnv_viterbi_cells on host matrices (pageable
    memory, like R's), without and with icnv_residency(1) (the HMM input is recognised and not uploaded again).
    Never the headline `value`: that is for matrices resident in HBM."""
    import ctypes as ct
    import numpy as np
    from infercnv_amd import _lib
    from infercnv_amd._lib import Cfg, check, f64, i32
    L = _lib.load()
    C = min(cells, x_dev.shape[0])
    G = x_dev.shape[1]
    x = x_dev[:C].cpu().numpy()                      # (C, G) row-major == genes x cells column-major
    refs_h = [r[r < C] for r in refs]
    out, pre = np.empty_like(x), np.empty_like(x)
    st = np.empty((C, G), dtype=np.uint8)
    means, sd, logPi, logDelta = hmm
    m, mp = f64(means)
    lp = np.asfortranarray(logPi)
    ld, ldp = f64(logDelta)
    csa, csp = i32(chr_start)
    vp = lambda a: a.ctypes.data_as(ct.c_void_p)
    cfg = Cfg(G, C, chr_start, refs_h)

    def run():
        check(L.icnv_smooth_chain(vp(x), vp(out), vp(pre), cfg.ptr()))
        check(L.icnv_viterbi_cells(vp(pre), vp(st), G, C, csp, csa.size - 1, len(means), mp, float(sd),
                                   lp.ctypes.data_as(ct.POINTER(ct.c_double)), ldp))

    res = {"cells": C, "what": "icnv_smooth_chain + icnv_viterbi_cells on host matrices (pageable, like R's), uploads and downloads included; "
                              "phases = icnv_host_path_stats of the timed pair of calls (wall-clock ms; with the pipeline the upload, kernel "
                              "and download phases overlap)"}
    names = ("calls", "fingerprint_ms", "hash_ms", "hash_threads", "h2d_ms", "h2d_bytes", "d2h_ms", "d2h_bytes", "device_ms", "pipelined_calls", "wall_ms", "alloc_ms")
    def timed(label, warm):
        # warm-up: pool blocks, tables, page faults of the output arrays.  With residency on, the matrices kept on the device
        # hold pool blocks until they are evicted (six per device): the pool reaches its steady state -- no hipMalloc inside a
        # call -- after three pairs of calls; rounds 3 and 4 timed the second pair and reported its allocations (172 ms)
        first = None
        for k in range(warm):
            L.icnv_host_path_stats_reset()
            t0 = time.perf_counter()
            run()
            if k == 1:
                first = (time.perf_counter() - t0) * 1e3
        L.icnv_host_path_stats_reset()
        t0 = time.perf_counter()
        run()
        t = time.perf_counter() - t0
        buf = (ct.c_double * 12)()
        check(L.icnv_host_path_stats(buf, 12))
        res[label] = {"value": C / t, "unit": "cells/s", "ms": t * 1e3, "warm_up_pairs_of_calls": warm,
                      "ms_of_the_second_pair_of_calls": first, "phases": {k: float(v) for k, v in zip(names, buf)}}
    os.environ.pop("ICNV_HOST_PIPELINE", None)
    check(L.icnv_residency(0))
    timed("residency_off", 2)                        # the default: three-thread pipeline over column blocks
    check(L.icnv_residency(1))
    timed("residency_on", 4)
    check(L.icnv_residency(0))
    res["default"] = "residency_off (also the R glue's default since round 5)"
    return res


def run_group_config(args, world, rank):
    """BASELINE configs 4 (i3 HMM at subcluster level) and 5 (2-D median filter): subclusters / tiles are whole on their
    rank (contiguous cell blocks cut at subcluster boundaries, sharded.align_to_groups); the inputs -- the smoothing
    chain's outputs -- are produced once, untimed; a step is one pass of the group HMM (two all-reduces of two doubles for
    the i3 mu / sigma, then rank-local) or of the median filter (no collective) over this rank's resident cells."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from infercnv_amd import device, sharded, synth
    G, C_total = args.genes, args.cells * world
    subs, is_ref, cuts = synth.subclusters(C_total)
    c0, c1 = sharded.align_to_groups(C_total, world, cuts)[rank]
    C_local = c1 - c0
    x, chr_start = synth.make_matrix_torch(G, C_local, "cuda", cell_offset=c0, C_total=C_total)
    refs_global, _ = synth.groups(C_total)
    plan = device.ChainPlan(G, C_local, chr_start, sharded.localize_groups(refs_global, c0, c1))
    out, pre = sharded.ShardedChain(plan).run(x, want_pre_denoise=True)
    del x
    mine = [i for i, g in enumerate(subs) if c0 <= int(g[0]) < c1]
    assert all(c0 <= int(subs[i].min()) and int(subs[i].max()) < c1 for i in mine), "a subcluster straddles two ranks"
    local = [(subs[i] - c0).astype(np.int32) for i in mine]
    ref_local = np.concatenate([local[j] for j, i in enumerate(mine) if is_ref[i]] or [np.zeros(0, dtype=np.int32)])
    dev = torch.device("cuda", torch.cuda.current_device())
    if args.config == 4:
        del out
        hmm = sharded.ShardedGroupHMM(device=dev)
        result = [None]
        def step():
            result[0] = hmm.run_i3(pre, chr_start, local, ref_local)
        names = ("cells_moments", "group_means", "viterbi", "viterbi_groups", "viterbi_redo", "viterbi_exact_fallback", "broadcast_states")
        what = "i3 HMM at subcluster level (R/inferCNV_i3HMM.R:249-308): i3 mu / sigma over the reference values (2 all-reduces of 2 doubles), group means, Viterbi per subcluster, broadcast"
        alg = 9 * G * C_local          # read every value once (group means), write one state byte per gene*cell
    else:
        mf = sharded.ShardedMedianFilter()
        result = [None]
        def step():
            result[0] = mf.run(out, chr_start, local, 7)
        def step_no_ties():      # the same filter on the matrix BEFORE step 22: no value repeats, no window has a majority value
            result[0] = mf.run(pre, chr_start, local, 7)
        names = ("median_filter",)
        what = "apply_median_filtering, window_size 7 (9 x 9 windows clamped at tile x chromosome edges; R/noise_reduction.R:43-113), tiles = subclusters x chromosomes, no collective"
        alg = 2 * 8 * G * C_local

    def fence():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    if not args.no_kernel_timing:
        device.timing_reset()
        device.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    device.timing_enable(False)
    no_ties_ms = None
    if args.config == 5 and not args.no_side_legs:
        step_no_ties()
        fence()
        t1 = time.perf_counter()
        for _ in range(3):
            step_no_ties()
        fence()
        no_ties_ms = (time.perf_counter() - t1) / 3 * 1e3
        step()                                  # (the checksum below is of the step's own output)
        fence()
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    cells = torch.tensor([float(C_local)], dtype=torch.float64, device="cuda")
    csum = torch.tensor([float(result[0].sum(dtype=torch.float64))], dtype=torch.float64, device="cuda")
    if dist.is_initialized():
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        allc = [torch.zeros_like(csum) for _ in range(world)]
        dist.all_gather(allc, csum)
        checks = [float(v.item()) for v in allc]
    else:
        checks = [float(csum.item())]
    elapsed = float(tmax.item())
    if rank == 0:
        kernels = {}
        for k in names:
            ms, n = device.timing_get(k)
            if n:
                kernels[k] = {"avg_ms": ms / n, "launches_per_step": n / args.steps, "ms_per_step": ms / args.steps}
        ms_per_step = elapsed / args.steps * 1e3
        ksum = sum(v["ms_per_step"] for v in kernels.values())
        res = {"metric": ("cells/sec through the i3 HMM at subcluster level, 10k genes" if args.config == 4 else
                          "cells/sec through apply_median_filtering, 10k genes"),
               "value": C_total * args.steps / elapsed, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic",
               "config": {"workload": f"BASELINE config {args.config}: synthetic {G} genes x {args.cells} cells per GPU ({C_total} total), "
                                      f"{what}; inputs (the smoothing chain's output) resident in HBM",
                          "genes": G, "cells_per_gpu": args.cells, "cells_total": C_total, "subclusters_rank0": len(local),
                          "parallelism": f"whole subclusters per GPU x{world} (contiguous blocks cut at subcluster boundaries)"},
               # priced over the STEP (what the job gets), not over the sum of the kernel times: the gap between the two is launch
               # gaps and host round trips, and it belongs to the step.  The kernel-sum figure sits under its own key.
               "roofline": {"bound": "hbm", "achieved": alg / (ms_per_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "frac_over_kernel_sum": (alg / (ksum * 1e-3) / 1e9 / HBM_PEAK_GBS) if kernels and ksum > 0 else None,
                            "ms_outside_kernels": (ms_per_step - ksum) if kernels else None,
                            "traffic": (_pmc_traffic().get("config%d_bytes_per_step" % args.config)
                                        if args.cells == 50000 and G == 10000 and world == 1 and _pmc_traffic().get("_current") else None),
                            **_traffic_fields(_pmc_traffic()),
                            "algorithmic_bytes_per_step": alg, "kernel_ms_per_step": ksum,
                            "note": "all kernels of the step together (group means / Viterbi / broadcast, or the interior and edge median kernels)"},
               "world": {"env_world_size": world, "n_gpus_arg": args.gpus,
                         "launcher": "bench.py itself (self_launch)" if os.environ.get("ICNV_BENCH_LAUNCHER") == "self" else "external (torchrun) or none",
                         "communicator_world_size": dist.get_world_size() if dist.is_initialized() else 1,
                         "backend": (dist.get_backend() if dist.is_initialized() else None)},
               "kernels": kernels, "cpu_baseline": None,
               **({"no_ties_input": {"ms_per_step": no_ties_ms, "frac": alg / (no_ties_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "what": "the same filter over the matrix BEFORE step 22 (no repeated values: no window has a majority value, "
                                             "every output runs its selection network) -- the data-independent floor of the kernels; the timed step "
                                             "runs on the denoised matrix, as BASELINE config 5 and apply_median_filtering's use specify, where "
                                             "the majority shortcut decides most windows"}} if no_ties_ms else {}),
               "checksums": {"per_rank": checks, "meaning": "sum of the step's output (states, or the filtered matrix) over the rank's cells"}}
        print(json.dumps(res))


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this very command line under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` (rendezvous on 127.0.0.1, a free port), one rank per
    GPU over RCCL, every flag passed through.  The ranks inherit stdout / stderr: rank 0 prints the one JSON line.  Returns
    the launcher's exit code.  (The `torchrun ... bench.py --gpus N` form keeps working: it sets WORLD_SIZE itself.)"""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, ICNV_BENCH_LAUNCHER="self")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs across processes on this driver
    env.setdefault("OMP_NUM_THREADS", "1")                 # (what torchrun would set, without its warning)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--genes", type=int, default=10000)
    ap.add_argument("--cells", type=int, default=50000, help="cells per GPU (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-side-legs", action="store_true",
                    help="only the warm-up and the timed steps (profiling runs: no extra launches on other data, e.g. config 5's no-ties input)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5),
                    help="BASELINE.json config: 2 (default) fused smooth chain + per-cell i6 HMM, 50 000 cells per GPU (weak scaling) -- the "
                         "headline metric; 3 the same step on BASELINE configs[2]: --total-cells (1 000 000) cells IN TOTAL dealt over the "
                         "N GPUs (strong scaling; N = 1 holds all of them: 270 GB of HBM); 4 i3 HMM at subcluster level; "
                         "5 apply_median_filtering (whole subclusters / tiles per GPU, SURVEY.md 8e)")
    ap.add_argument("--total-cells", type=int, default=1000000, help="config 3: cells of the whole job")
    ap.add_argument("--checksum", type=int, default=0, metavar="PARTS",
                    help="also print checksums of the outputs: per rank (N > 1), or -- on one rank -- per residue class of the cell "
                         "index modulo PARTS, i.e. the cells rank r of a PARTS-rank run holds (tests/test_gpu_entrypoints.py)")
    ap.add_argument("--dump", default=None, metavar="DIR",
                    help="with --checksum: also write, per rank / per residue class r, DIR/states_r.npy (uint8 state calls), "
                         "DIR/pre_cellsums_r.npy and DIR/out_cellsums_r.npy (per-cell sums of the HMM input and of the denoised matrix): "
                         "what tests/test_gpu_entrypoints.py compares element by element between an N-rank and a one-rank run")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: no launcher has set the rank environment, so this process becomes the launcher
        raise SystemExit(self_launch(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist
    from infercnv_amd import device, sharded, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher's WORLD_SIZE is {world}: launch `python bench.py --gpus N` by itself "
                         "(it starts its own ranks) or under torch.distributed.run --nproc-per-node N")
    if os.environ.get("ICNV_BENCH_ONE_DEVICE"):          # smoke-test the N>1 code path on a single GPU
        local_rank = 0
    local_rank %= max(torch.cuda.device_count(), 1)      # (a launcher that shows every rank only its own GPU: that one is device 0)
    torch.cuda.set_device(local_rank)
    device.init(local_rank)
    # ICNV_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, barriers, all-reduces, all-gathers) with ONE rank --
    # the only way to run it over RCCL on a single-GPU box (tests/test_gpu_entrypoints.py)
    dist_on = world > 1 or bool(os.environ.get("ICNV_BENCH_FORCE_DIST"))
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29591")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("ICNV_BENCH_BACKEND", "nccl")   # "gloo" only for the 1-GPU smoke of the N>1 path
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    global STREAM_1R2W_GBS, STREAM_1R2W_SOURCE, COLUMN_WALK_GBS, COLUMN_WALK_SOURCE, ROW_REQUESTS_GBS, ROW_REQUESTS_SOURCE
    if rank == 0 and args.config in (2, 3) and not os.environ.get("ICNV_BENCH_NO_UBENCH"):
        ceil = measure_ceilings()                    # on this box, in this run, before any bench tensor exists
        STREAM_1R2W_GBS, STREAM_1R2W_SOURCE = ceil["stream_1r2w_gbs"], ceil["stream_1r2w_source"]
        COLUMN_WALK_GBS, COLUMN_WALK_SOURCE = ceil["column_walk_gbs"], ceil["column_walk_source"]
        ROW_REQUESTS_GBS, ROW_REQUESTS_SOURCE = ceil.get("row_requests_gbs"), ceil.get("row_requests_source")
    if dist_on:
        dist.barrier()

    G = args.genes
    if args.config == 3:                             # strong scaling: the job's cells dealt round-robin over the ranks
        C_total = args.total_cells
        C_local = len(range(rank, C_total, world))
    else:
        C_local = args.cells
        C_total = C_local * world
    if args.config in (4, 5):
        run_group_config(args, world, rank)
        if dist_on:
            dist.barrier()
            dist.destroy_process_group()
        return
    # cells are dealt to the ranks round-robin (sharded.cyclic_cells): every rank holds its share of every
    # reference group, so the reference rounds are balanced (the generator puts the reference cells first)
    x, chr_start = synth.make_matrix_torch(G, C_local, "cuda", cell_offset=rank, cell_stride=world, C_total=C_total)
    refs_global, _ = synth.groups(C_total)
    refs_local = sharded.localize_groups_cyclic(refs_global, rank, world)
    means, sd, logPi, logDelta = synth.hmm_params_i6()

    out = torch.empty_like(x)
    pre_buf = torch.empty_like(x)                    # the HMM input (the matrix before step 22), written by the apply pass
    states = torch.empty((C_local, G), dtype=torch.uint8, device="cuda")
    plan = device.ChainPlan(G, C_local, chr_start, refs_local)
    chain = sharded.ShardedChain(plan, always_reduce=dist_on)

    def step():
        _, pre = chain.run(x, out=out, pre=pre_buf)
        device.viterbi_cells(pre, chr_start, means, sd, logPi, logDelta, states=states)
        return pre

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # Inside the timed region only the two hot launches carry HIP events (level 2): an event pair costs a few microseconds of
    # stream time, and with all ~14 launches of a step bracketed the step is 2.5 % slower (5.27 against 5.13 ms, measured).
    # The other kernel families are timed in a short DETAIL pass right behind the timed region (level 1, same state).
    if not args.no_kernel_timing:
        device.timing_reset()
        device.timing_enable(2)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    device.timing_enable(False)
    hot = {}
    if not args.no_kernel_timing:
        for k in ("chain_apply", "viterbi"):
            ms, n = device.timing_get(k)
            if n:
                hot[k] = {"avg_ms": ms / n, "launches_per_step": n / args.steps, "ms_per_step": ms / args.steps,
                          "measured": "HIP events on the launch stream inside the timed region"}
    detail_steps, detail_ms = 0, None
    if not args.no_kernel_timing:
        detail_steps = max(1, min(5, args.steps))
        device.timing_reset()
        device.timing_enable(1)
        fence()
        td = time.perf_counter()
        for _ in range(detail_steps):
            step()
        fence()
        detail_ms = (time.perf_counter() - td) / detail_steps * 1e3
        device.timing_enable(False)

    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    cells_per_rank = [C_local]
    if dist_on:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        mine = torch.tensor([float(C_local)], dtype=torch.float64, device="cuda")
        allc = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(allc, mine)
        cells_per_rank = [int(v.item()) for v in allc]
    elapsed = float(tmax.item())

    checksums = None
    if args.checksum:
        pre_last = step()
        torch.cuda.synchronize()
        def sums(sel):
            return [float(out[sel].sum(dtype=torch.float64)), float(pre_last[sel].sum(dtype=torch.float64)),
                    int(states[sel].sum(dtype=torch.int64))]
        if dist_on and not (world == 1 and args.checksum > 1):
            mine = torch.tensor(sums(slice(None)), dtype=torch.float64, device="cuda")
            allv = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allv, mine)
            checksums = [[float(v[0]), float(v[1]), int(v[2])] for v in allv]
        else:
            checksums = [sums(slice(r, None, args.checksum)) for r in range(args.checksum)]
        if args.dump:
            os.makedirs(args.dump, exist_ok=True)
            def dump(sel, r):
                np.save(os.path.join(args.dump, f"states_{r}.npy"), states[sel].cpu().numpy())
                np.save(os.path.join(args.dump, f"pre_cellsums_{r}.npy"), pre_last[sel].sum(dim=1, dtype=torch.float64).cpu().numpy())
                np.save(os.path.join(args.dump, f"out_cellsums_{r}.npy"), out[sel].sum(dim=1, dtype=torch.float64).cpu().numpy())
            if dist_on and not (world == 1 and args.checksum > 1):
                dump(slice(None), rank)
            else:
                for r in range(args.checksum):
                    dump(slice(r, None, args.checksum), r)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = C_total * args.steps / elapsed
        kernels = {}
        for k in ("chain_apply", "chain_apply_ref", "chain_stage_ref", "chain_gene_sums", "chain_cell_stats", "viterbi", "viterbi_redo", "reduce_partials"):
            ms, n = device.timing_get(k)      # the detail pass (every family bracketed by events)
            if n:
                kernels[k] = {"avg_ms": ms / n, "launches_per_step": n / detail_steps, "ms_per_step": ms / detail_steps,
                              "measured": f"HIP events, detail pass of {detail_steps} steps right behind the timed region"}
        detail = {k: dict(v) for k, v in kernels.items()}
        kernels.update(hot)                   # the two hot launches: from the timed region itself
        # algorithmic bytes per launch: chain apply reads 8 B and writes 8 B per gene*cell
        # (+8 B for the HMM-input copy it also emits here); Viterbi reads 8 B and writes 1 B per gene*cell.
        # With the reference-cell cache the dominant chain_apply launch covers the non-reference cells only;
        # the reference cells continue from the cache in chain_apply_ref (elementwise, 8 B in + 16 B out).
        n_ref_local = int(sum(len(g) for g in refs_local))
        n_main = C_local - n_ref_local if "chain_apply_ref" in kernels else C_local
        alg = {"chain_apply": 2 * 8 * G * n_main, "viterbi": 9 * G * C_local}
        if "chain_apply_ref" in kernels:
            alg["chain_apply_ref"] = 2 * 8 * G * n_ref_local
        roof = {}
        for k, b in alg.items():
            if k in kernels:
                # (a family with several launches per step -- the Viterbi's column batches past 429 000 cells -- is priced per
                # launch: its bytes per step / launches per step over its average launch duration = bytes per step / ms per step)
                gbs = b / (kernels[k]["ms_per_step"] * 1e-3) / 1e9
                roof[k] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                           "algorithmic_bytes_per_launch": b / max(kernels[k]["launches_per_step"], 1e-9),
                           "launches_per_step": kernels[k]["launches_per_step"], "avg_launch_ms": kernels[k]["avg_ms"]}
        if "chain_apply" in roof:
            cap = STREAM_1R2W_GBS / HBM_PEAK_GBS * 16.0 / 24.0
            roof["chain_apply"]["structural_cap_of_frac"] = cap
            roof["chain_apply"]["frac_of_structural_cap"] = roof["chain_apply"]["frac"] / cap
            roof["chain_apply"]["note"] = (
                f"fused smooth pass (steps 8-14 + 22) over the {n_main} non-reference cells of this rank: reads each cell once, writes "
                "the denoised matrix and -- not counted in the algorithmic bytes -- the pre-denoise HMM input (+8 B/gene*cell, the "
                "Viterbi's observations, which infercnv::run() needs as a matrix of its own): 24 B of HBM traffic per gene*cell for "
                "16 algorithmic.  `frac` prices the 16 algorithmic bytes against the 8 TB/s spec.  STRUCTURAL CAP: a plain "
                f"1-read : 2-write stream reaches {STREAM_1R2W_GBS:.0f} GB/s on this GPU ({STREAM_1R2W_SOURCE}), so with two output "
                f"matrices `frac` cannot exceed {STREAM_1R2W_GBS:.0f} / 8000 x 16 / 24 = {cap:.2f}; the north star's 0.70 is out of reach for "
                f"the pass as specified, and this launch sits at {roof['chain_apply']['frac'] / cap:.2f} of the cap (`hbm_traffic` prices the "
                "bytes really moved).  What paces it (DESIGN.md K2): ~900 vector instructions per thread and cell -- issue for more than half "
                "of the ~28 000 cycles a CU spends on a cell, the rest barrier-separated latency; across boxes whose stream rates differ by "
                "15 % the launch time moves by 2 % (round 5), confined to half the CUs it takes 1.73 x as long for twice the cells per CU "
                "(round 4, profiles/r04_cu_mask_probe.txt)")
        if "viterbi" in roof:
            st = device.viterbi_last_stats()
            roof["viterbi"]["note"] = (f"certified fast path ({st['path']}, {st.get('kernel', '?')} kernel, {st['table_intervals']} table records; "
                                       f"{st['flagged']} of {st['sequences']} sequences redone exactly): "
                                       "table-driven emission scores (degree-4 polynomials on a uniform grid) + max-plus recurrence, one lane per "
                                       "sequence, 768 threads (three wavefronts per SIMD), the gene step software-pipelined (gathers | decision "
                                       "bookkeeping | rows), 84 vector + 15 LDS + 18 scalar + 1 store instructions per gene and wavefront in "
                                       "the forward pass, block summaries for the traceback.  Round 5 (DESIGN.md K4b): the STAGED kernel requests the "
                                       "observations by whole cache lines -- eight 128-byte lines per request, LDS-DMA, every lane then reads its own "
                                       "column out of the wavefront's 8 KiB buffer -- instead of every lane walking a column of its own: "
                                       "`row_requests_ceiling` against `column_walk_ceiling` are the two patterns without arithmetic, both measured "
                                       "in this run; with the observations served from L2 the 50 000-cell launch took 1.73 ms (round-4 ablation): "
                                       "the launch is now paced by vector issue.  No MFMA-shaped work")
        if "viterbi" in roof:
            # second ceiling of the Viterbi (SURVEY.md 8d asks for HBM GB/s *and* the fp64 rate): vector instructions per gene and
            # wavefront (SQ_INSTS_VALU of the launch / gene steps, profiles/r06_pmc_viterbi_fast.txt; 84 of them in the forward pass
            # by static count, scripts/vf_asm_stats.py), every one of them 4 cycles of a 16-lane SIMD; 256 CUs x 4 SIMDs at the 2.4 GHz peak clock
            instr = VITERBI_VALU_PER_GENE
            ceil_ms = (G * C_local / 64.0) * instr * 4.0 / (256 * 4) / 2.4e9 * 1e3
            roof["viterbi"]["fp64_issue"] = {"vector_instr_per_gene_wavefront": instr, "ceiling_ms": ceil_ms,
                                             "frac": ceil_ms / kernels["viterbi"]["ms_per_step"],
                                             "fp64_vector_peak_tflops": FP64_VECTOR_PEAK_TF,
                                             "note": "share of the launch the vector pipes would need at full issue rate and the 2.4 GHz peak "
                                                     "clock (the chip holds ~2.05-2.1 GHz under this load)"}
            roof["viterbi"]["column_walk_ceiling"] = {"gbs": COLUMN_WALK_GBS, "source": COLUMN_WALK_SOURCE,
                                                      "frac_of_column_walk": 8.0 * G * C_local / COLUMN_WALK_GBS / 1e6 / kernels["viterbi"]["ms_per_step"],
                                                      "ms_for_the_observations_alone": 8.0 * G * C_local / COLUMN_WALK_GBS / 1e6,
                                                      "note": "the register kernel's pattern (icnv_viterbi_set_mode(2); batches that leave the staged kernel's table)"}
            if ROW_REQUESTS_GBS:
                roof["viterbi"]["row_requests_ceiling"] = {"gbs": ROW_REQUESTS_GBS, "source": ROW_REQUESTS_SOURCE,
                                                           "ms_for_the_observations_alone": 8.0 * G * C_local / ROW_REQUESTS_GBS / 1e6,
                                                           "note": "the staged kernel's pattern"}
        dominant = max(kernels, key=lambda k: kernels[k]["ms_per_step"]) if kernels else None
        if "chain_apply" in roof:
            moved = 3 * 8 * G * n_main            # one matrix read, two written (refined below by the counters when present)
            t = kernels["chain_apply"]["ms_per_step"] * 1e-3
            roof["chain_apply"]["hbm_traffic"] = {"bytes_per_launch": moved, "source": "3 x 8 B per gene*cell",
                                                  "achieved": moved / t / 1e9, "unit": "GB/s",
                                                  "frac_of_peak": moved / t / 1e9 / HBM_PEAK_GBS,
                                                  "stream_1r2w": STREAM_1R2W_GBS, "stream_1r2w_source": STREAM_1R2W_SOURCE,
                                                  "frac_of_stream_1r2w": moved / t / 1e9 / STREAM_1R2W_GBS}
        tr = _pmc_traffic()
        if tr and G == 10000 and C_local == 50000:   # counters were collected on this shape
            for k in roof:
                if k in tr:
                    roof[k].update(_traffic_fields(tr))
                    if not tr.get("_current"):
                        continue                         # counters of other sources: the line says so and quotes no traffic
                    roof[k]["traffic"] = tr[k]
                    if k == "chain_apply":
                        ht, t = roof[k]["hbm_traffic"], kernels[k]["avg_ms"] * 1e-3
                        ht.update({"bytes_per_launch": tr[k], "source": "FETCH_SIZE (x2) + WRITE_SIZE counters, see traffic_source",
                                   "achieved": tr[k] / t / 1e9, "frac_of_peak": tr[k] / t / 1e9 / HBM_PEAK_GBS,
                                   "frac_of_stream_1r2w": tr[k] / t / 1e9 / STREAM_1R2W_GBS})
        if args.config == 3:
            workload = (f"BASELINE configs[2]: synthetic {G} genes x {C_total} cells IN TOTAL, dealt round-robin over {world} GPU(s) "
                        f"({C_local} on rank 0), fused smooth chain (steps 8,9,10,11,12,14,22) + per-cell i6 HMM Viterbi, inputs resident in HBM")
        else:
            workload = (f"BASELINE configs[1] per GPU: synthetic {G} genes x {C_local} cells per GPU ({C_total} total), fused smooth chain "
                        "(steps 8,9,10,11,12,14,22) + per-cell i6 HMM Viterbi, inputs resident in HBM")
        res = {
            "metric": "cells/sec through smooth+i6-HMM, 10k genes", "value": value, "unit": "cells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if args.config == 3 else "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "baseline_config": args.config,
                       "genes": G, "cells_per_gpu": C_local, "cells_total": C_total, "window_length": 101,
                       "hmm": "i6, t=1e-6", "parallelism": f"cell-shard x{world} (round-robin deal), 3 small all-reduces"},
            # what the driver's SCALE record can be checked against: the launcher's world size, the communicator's own, the
            # backend, and the cells every rank really held (all-gathered)
            "world": {"env_world_size": world, "n_gpus_arg": args.gpus,
                      "launcher": "bench.py itself (self_launch)" if os.environ.get("ICNV_BENCH_LAUNCHER") == "self" else "external (torchrun) or none",
                      "communicator_world_size": dist.get_world_size() if dist_on else 1,
                      "backend": (dist.get_backend() if dist_on else None),
                      "cells_per_rank": cells_per_rank, "cells_sum": int(sum(cells_per_rank))},
            "ceilings_measured_in_this_run": {"stream_1r2w_gbs": STREAM_1R2W_GBS, "stream_1r2w_source": STREAM_1R2W_SOURCE,
                                              "column_walk_gbs": COLUMN_WALK_GBS, "column_walk_source": COLUMN_WALK_SOURCE,
                                              "row_requests_gbs": ROW_REQUESTS_GBS, "row_requests_source": ROW_REQUESTS_SOURCE},
            "roofline": roof.get(dominant) or (next(iter(roof.values())) if roof else None),
            "roofline_kernel": dominant,
            # the north star states its roofline target on the fused smooth pass: always there, whichever kernel is the
            # largest on this box (the pass and the Viterbi are within 5 % of each other)
            "roofline_fused_smooth_pass": roof.get("chain_apply"),
            "roofline_by_kernel": roof,
            # the whole step against the same roofline: algorithmic bytes of the pass (16 B per gene*cell) and of the Viterbi (9 B)
            "roofline_step": {"bound": "hbm", "algorithmic_bytes_per_step": (16 + 9) * G * C_local,
                              "achieved": (16 + 9) * G * C_local / (ms_per_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": (16 + 9) * G * C_local / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "ms_outside_kernels": (detail_ms - sum(v["ms_per_step"] for v in detail.values())) if detail else None,
                              "ms_outside_kernels_note": "of the detail pass: its own step time minus its own kernel times"},
            "kernels": kernels,
            "kernel_timing": {"timed_region": "events on chain_apply and viterbi only (icnv_timing_enable(2))",
                              "detail_pass": {"steps": detail_steps, "ms_per_step": detail_ms,
                                              "hot_kernels_there": {k: detail[k]["avg_ms"] for k in ("chain_apply", "viterbi") if k in detail}}},
        }
        if checksums is not None:
            res["checksums"] = {"per_part": checksums, "meaning": "[sum(denoised), sum(hmm_input), sum(states)] of the cells of rank r"}
        if not args.no_cpu_baseline and world == 1 and args.config == 3:
            # the oracle runs over [all reference cells | sampled observation cells] of THIS matrix: timed, it is the
            # cpu_baseline sample; compared with the step's outputs, it is the parity block
            res["cpu_baseline"], res["parity"] = cpu_baseline_on_matrix(x, out, pre_buf, states, chr_start, refs_local,
                                                                        (means, sd, logPi, logDelta))
        elif not args.no_cpu_baseline and world == 1:
            try:
                res["host_path"] = host_path_rate(x, chr_start, refs_local, (means, sd, logPi, logDelta))
            except Exception as e:          # a reported side figure must not take the bench line down
                res["host_path"] = {"error": str(e)[:200]}
            del x, out, pre_buf, states, chain, plan        # the bench tensors: the parity leg below holds its own sample in HBM
            res["cpu_baseline"], res["parity"] = cpu_baseline(G)
        elif not args.no_cpu_baseline:
            res["cpu_baseline"] = None
        print(json.dumps(res))

    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
