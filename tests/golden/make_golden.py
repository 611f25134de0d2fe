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
  data/mcmc_obj.rda                 -- realistic i6 emission means / precisions.
  inst/extdata/gencode_downsampled.EXAMPLE_ONLY_DONT_REUSE.txt -- genes per chr
      used to shape the synthetic benchmark.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import rda  # noqa: E402

REF = "/root/reference"


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


if __name__ == "__main__":
    main()
