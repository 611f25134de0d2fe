#!/usr/bin/env python3
"""Extract the reference's own golden data into small fixtures (run in the
build container, where /root/reference exists; the GPU box only sees the
committed outputs).

  python tests/golden/make_golden.py

Sources (all under /root/reference):
  data/infercnv_object_example.rda  -- @count.data -> @expr.data of a real
      infercnv::run(cutoff=1, cluster_by_groups=TRUE, HMM=FALSE, denoise=TRUE)
      (R/data.R:23-28): the known-answer test for steps 3,4,8,9,10,11,12,14,22.
  data/HMM_states.rda               -- i6 states of a `samples`-mode run
      (R/data.R:30-35); emission parameters were RNG-derived, so this is a
      sanity fixture only.
  data/mcmc_obj.rda                 -- the i6 emission means / precisions of the run that produced HMM_states.rda
      (@mu, @sig: that run's get_spike_dists output, R/inferCNV_BayesNet.R:154-161) and @cell_gene / @cnv_regions: the
      nine CNV regions (name, gene rows, cell columns) the reference derived from HMM_states.rda -> mcmc_cell_gene.npz.
  inst/extdata/gencode_downsampled.EXAMPLE_ONLY_DONT_REUSE.txt -- genes per chr
      used to shape the synthetic benchmark.
  inst/extdata/oligodendroglioma_{expression_downsampled.counts.matrix.gz,
      annotations_downsampled.txt} + the gene-position file -- the inputs of
      example/run.R (BASELINE.json configs[0]): CreateInfercnvObject
      (R/inferCNV.R:133-345) replayed here -> example_run_inputs.npz (the object
      as run() receives it: ordered count matrix, gene_order, reference /
      observation groups).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import rda  # noqa: E402

REF = "/root/reference"


def example_run_inputs():
    """CreateInfercnvObject (R/inferCNV.R:133-345) on example/run.R's inputs, with its defaults
    chr_exclude = c('chrX','chrY','chrM'), min_max_counts_per_cell = c(100, +Inf), max_cells_per_group = NULL and
    ref_group_names = c("Microglia/Macrophage", "Oligodendrocytes (non-malignant)") (example/run.R:8-12).
    The matrix values are decimals with at most 6 significant digits: stored as integer mantissa / power of ten
    (value = mant / 10**neg_exp, exactly the double read.table() parses), which compresses to half of the doubles."""
    import gzip
    from decimal import Decimal
    ext = f"{REF}/inst/extdata"
    with gzip.open(f"{ext}/oligodendroglioma_expression_downsampled.counts.matrix.gz", "rt") as fh:
        cells = fh.readline().rstrip("\n").split("\t")
        genes, rows = [], []
        for line in fh:
            p = line.rstrip("\n").split("\t")
            genes.append(p[0])
            rows.append(p[1:])
    genes = np.array(genes)
    mant = np.zeros((len(genes), len(cells)), dtype=np.int64)
    nexp = np.zeros((len(genes), len(cells)), dtype=np.int8)
    for i, r in enumerate(rows):
        for j, tok in enumerate(r):
            if tok == "0":
                continue
            sign, digits, e = Decimal(tok).as_tuple()
            assert sign == 0 and e <= 0 and -e <= 22
            m = int("".join(map(str, digits)))
            assert m < 2 ** 31
            mant[i, j], nexp[i, j] = m, -e
            assert m / 10.0 ** (-e) == float(tok)          # one correctly rounded division == strtod
    # gene positions; chr_exclude (:189-191)
    pos_name, pos_chr, pos_start, pos_stop = [], [], [], []
    with open(f"{ext}/gencode_downsampled.EXAMPLE_ONLY_DONT_REUSE.txt") as fh:
        for line in fh:
            g, c, a, b = line.rstrip("\n").split("\t")
            if c in ("chrX", "chrY", "chrM"):
                continue
            pos_name.append(g); pos_chr.append(c); pos_start.append(int(a)); pos_stop.append(int(b))
    assert len(set(pos_name)) == len(pos_name)
    # .order_reduce (R/inferCNV.R:352-428): drop start+stop == 0, keep genes present in both (matrix order), chr levels
    # in order of first appearance in the position table, stable order(chr, start, stop)
    ok = [i for i in range(len(pos_name)) if pos_start[i] + pos_stop[i] != 0]
    chr_levels = list(dict.fromkeys(pos_chr[i] for i in ok))
    where = {pos_name[i]: i for i in ok}
    keep = [gi for gi, g in enumerate(genes) if g in where]
    key = sorted(range(len(keep)), key=lambda k: (chr_levels.index(pos_chr[where[genes[keep[k]]]]),
                                                 pos_start[where[genes[keep[k]]]], pos_stop[where[genes[keep[k]]]]))
    gene_rows = np.array([keep[k] for k in key])
    chr_codes = np.array([chr_levels.index(pos_chr[where[genes[g]]]) for g in gene_rows], dtype=np.int32)
    used = sorted(set(chr_codes.tolist()))                 # droplevels (:240)
    chr_levels = [chr_levels[c] for c in used]
    chr_codes = np.searchsorted(np.array(used), chr_codes).astype(np.int32)
    mant, nexp = mant[gene_rows], nexp[gene_rows]
    x = mant / 10.0 ** nexp.astype(np.float64)
    # cells: colSums filter (:253-266), annotated cells only, classifications in matrix column order (:290-300)
    cs = x.sum(axis=0)
    keep_c = np.nonzero((cs >= 100) & (cs <= np.inf))[0]
    annot = {}
    with open(f"{ext}/oligodendroglioma_annotations_downsampled.txt") as fh:
        for line in fh:
            c, a = line.rstrip("\n").split("\t")
            annot[c] = a
    assert all(c in cells for c in annot)
    keep_c = np.array([j for j in keep_c if cells[j] in annot])
    mant, nexp = mant[:, keep_c], nexp[:, keep_c]
    cls = np.array([annot[cells[j]] for j in keep_c])
    ref_names = ["Microglia/Macrophage", "Oligodendrocytes (non-malignant)"]
    obs_names = sorted(set(cls) - set(ref_names))
    out = dict(mant=mant.astype(np.int32), neg_exp=nexp, chr_codes=chr_codes, chr_levels=np.array(chr_levels),
               ref_names=np.array(ref_names), obs_names=np.array(obs_names))
    for i, n in enumerate(ref_names):
        out[f"ref_{i}"] = np.nonzero(cls == n)[0].astype(np.int32)
    for i, n in enumerate(obs_names):
        out[f"obs_{i}"] = np.nonzero(cls == n)[0].astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "example_run_inputs.npz"), **out)
    print("example/run.R inputs:", mant.shape, "chr", len(chr_levels), "refs", [len(out[f"ref_{i}"]) for i in range(2)],
          "obs", {n: len(out[f"obs_{i}"]) for i, n in enumerate(obs_names)})


def mcmc_cell_gene(mc=None):
    """data/mcmc_obj.rda @cell_gene / @cnv_regions -> mcmc_cell_gene.npz: the nine CNV regions the reference itself derived
    from data/HMM_states.rda (man/filterHighPNormals.Rd:29-31 uses the two objects as a pair) -- generate_cnv_region_reports
    (R/inferCNV_HMM.R:790-869: .get_state_consensus, .define_cnv_gene_regions :1005-1057, ignore_neutral_state = 3) wrote
    17_HMM_pred*.pred_cnv_genes.dat / .cell_groupings, initializeObject / getGenesCells (R/inferCNV_BayesNet.R:245-316) read
    them back: region names in order of appearance, gene rows and cell columns as 1-BASED indices (kept 1-based here, as
    stored), the factor's levels (sorted names) and codes.  A reference-held golden for SURVEY.md 8f #2."""
    if mc is None:
        mc = rda.read_rda(f"{REF}/data/mcmc_obj.rda")["mcmc_obj"]
    cr = mc.attrs["cnv_regions"]
    levels = [str(v) for v in cr.attrs["levels"]]
    codes = np.asarray(cr.value, dtype=np.int32)
    out = dict(levels=np.array(levels), codes=codes, names=np.array([levels[c - 1] for c in codes]),
               group_id=np.asarray(mc.attrs["group_id"], dtype=np.int64),
               obs_tumor=np.asarray(mc.attrs["observation_grouped_cell_indices"].value["tumor"], dtype=np.int32),
               ref_normal=np.asarray(mc.attrs["reference_grouped_cell_indices"].value["normal"], dtype=np.int32))
    for i, e in enumerate(mc.attrs["cell_gene"]):
        v = e.value
        f = v["cnv_regions"]
        f = f[0] if isinstance(f, (list, np.ndarray)) else f
        name = [str(x) for x in f.attrs["levels"]][int(np.asarray(f.value).ravel()[0]) - 1]
        assert name == out["names"][i]
        out[f"genes_{i}"] = np.asarray(v["Genes"], dtype=np.int32)
        out[f"cells_{i}"] = np.asarray(v["Cells"], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "mcmc_cell_gene.npz"), **out)
    print("mcmc_obj@cell_gene:", [(str(n), len(out[f"genes_{i}"]), len(out[f"cells_{i}"])) for i, n in enumerate(out["names"])])


STEP_FUNCTIONS = {   # the step functions the R glue re-binds (rglue/R/zzz_hip_backend.R: .icnv_enable_hip_backend), and their files
    "R/inferCNV_ops.R": ["subtract_ref_expr_from_obs", "apply_max_threshold_bounds", "smooth_by_chromosome",
                         "center_cell_expr_across_chromosome", "invert_log2", "clear_noise_via_ref_mean_sd", "clear_noise",
                         "get_average_bounds", "scale_infercnv_expr", "remove_outliers_norm"],
    "R/inferCNV_HMM.R": ["predict_CNV_via_HMM_on_indiv_cells", "predict_CNV_via_HMM_on_tumor_subclusters",
                         "predict_CNV_via_HMM_on_tumor_subclusters_per_chr", "predict_CNV_via_HMM_on_whole_tumor_samples",
                         "assign_HMM_states_to_proxy_expr_vals"],
    "R/inferCNV_i3HMM.R": ["i3HMM_predict_CNV_via_HMM_on_indiv_cells", "i3HMM_predict_CNV_via_HMM_on_tumor_subclusters",
                           "i3HMM_predict_CNV_via_HMM_on_whole_tumor_samples", "i3HMM_assign_HMM_states_to_proxy_expr_vals"],
    "R/noise_reduction.R": ["apply_median_filtering"],
}


def step_function_signatures():
    """Names, order and default-carrying flags of the formals of the reference's step functions ->
    r_step_function_signatures.json: what the R glue's replacements must accept (tests/test_host.py)."""
    import json
    sys.path.insert(0, HERE)
    import r_signatures
    out = {}
    for f, names in STEP_FUNCTIONS.items():
        src = open(f"{REF}/{f}").read()
        for n in names:
            fm = r_signatures.formals(src, n)
            assert fm, (f, n)
            out[n] = {"file": f, "formals": [a for a, _ in fm], "has_default": [bool(d) for _, d in fm]}
    with open(os.path.join(HERE, "r_step_function_signatures.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("step function signatures:", {k: v["formals"] for k, v in out.items()})


def main():
    obj = rda.read_rda(f"{REF}/data/infercnv_object_example.rda")["infercnv_object_example"]
    expr = rda.as_matrix(obj.attrs["expr.data"])
    counts = rda.as_matrix(obj.attrs["count.data"]).astype(np.int32)
    go = obj.attrs["gene_order"]
    chr_codes, chr_levels = rda.factor_codes(go.value["chr"])
    ref = {k: np.asarray(v, dtype=np.int32) - 1 for k, v in obj.attrs["reference_grouped_cell_indices"].value.items()}
    obs = {k: np.asarray(v, dtype=np.int32) - 1 for k, v in obj.attrs["observation_grouped_cell_indices"].value.items()}
    sub = obj.attrs["tumor_subclusters"].value["subclusters"].value
    subclusters = {}
    for grp, d in sub.items():
        d = d.value if hasattr(d, "value") else d
        for name, v in d.items():
            v = v.value if hasattr(v, "value") else v   # named int vector
            subclusters[f"{grp}/{name}"] = np.asarray(v, dtype=np.int32) - 1
    np.savez_compressed(
        os.path.join(HERE, "infercnv_object_example.npz"),
        expr_data=expr, count_data=counts,
        chr_codes=chr_codes.astype(np.int32), chr_levels=np.array(chr_levels),
        gene_start=np.asarray(go.value["start"]), gene_stop=np.asarray(go.value["stop"]),
        ref_normal=ref["normal"], obs_tumor=obs["tumor"],
        subcluster_names=np.array(list(subclusters)),
        **{f"subcluster_{i}": v for i, v in enumerate(subclusters.values())},
    )

    hs = rda.as_matrix(rda.read_rda(f"{REF}/data/HMM_states.rda")["HMM_states"]).astype(np.int8)
    mc = rda.read_rda(f"{REF}/data/mcmc_obj.rda")["mcmc_obj"]
    np.savez_compressed(os.path.join(HERE, "hmm_states_example.npz"), HMM_states=hs,
                        mu=np.asarray(mc.attrs["mu"]), sig=np.asarray(mc.attrs["sig"]))

    mcmc_cell_gene(mc)
    step_function_signatures()

    # genes per chromosome of the bundled gene-position file, file order
    counts_by_chr = {}
    with open(f"{REF}/inst/extdata/gencode_downsampled.EXAMPLE_ONLY_DONT_REUSE.txt") as fh:
        for line in fh:
            c = line.split("\t")[1]
            counts_by_chr[c] = counts_by_chr.get(c, 0) + 1
    with open(os.path.join(HERE, "gencode_genes_per_chr.txt"), "w") as fh:
        for c, n in counts_by_chr.items():
            fh.write(f"{c}\t{n}\n")
    print("expr", expr.shape, "counts", counts.shape, "chr levels", len(chr_levels),
          "HMM_states", hs.shape, "chr counts", counts_by_chr)
    example_run_inputs()


if __name__ == "__main__":
    main()
