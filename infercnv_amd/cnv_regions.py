"""CNV region / consensus reporting (second "next" row): host-side mirror of
get_predicted_CNV_regions, .define_cnv_gene_regions, .get_cnv_gene_region_bounds and
generate_cnv_region_reports (R/inferCNV_HMM.R:706-869, 1005-1087).  The per-gene consensus over a
group's cells (.get_state_consensus, :977-987) runs on the GPU (icnv_state_consensus); the run-length
segmentation and the four report files are tiny and stay on the host, in the reference's formats.
"""
from __future__ import annotations

import ctypes as ct
import os

import numpy as np

from . import _lib
from ._lib import check, i32, pack_groups
from .infercnv_object import InfercnvObject


def _states_u8(obj):
    st = np.asarray(obj.expr_data)
    return np.asfortranarray(np.where(st < 0, 255, st).astype(np.uint8))


def state_consensus(infercnv_obj: InfercnvObject, groups):
    """(G, n_groups) consensus states (float, -1 where the consensus is the invalid state)."""
    L = _lib.load()
    st = _states_u8(infercnv_obj)
    G, C = st.shape
    idx, off = pack_groups(groups)
    idx, ip = i32(idx)
    off, op = i32(off)
    cons = np.empty((G, len(groups)), dtype=np.uint8, order="F")
    check(L.icnv_state_consensus(st.ctypes.data_as(ct.c_void_p), G, C, ip, op, len(groups),
                                 cons.ctypes.data_as(ct.c_void_p), None))
    out = cons.astype(np.float64)
    out[cons == 255] = -1.0
    return out


def overwrite_with_consensus(infercnv_obj: InfercnvObject, groups) -> InfercnvObject:
    """expr.data[genes, group_cells] <- consensus state, the effect of R/inferCNV_HMM.R:473-483."""
    L = _lib.load()
    st = _states_u8(infercnv_obj)
    G, C = st.shape
    idx, off = pack_groups(groups)
    idx, ip = i32(idx)
    off, op = i32(off)
    out = np.empty((G, C), dtype=np.uint8, order="F")
    check(L.icnv_state_consensus(st.ctypes.data_as(ct.c_void_p), G, C, ip, op, len(groups), None,
                                 out.ctypes.data_as(ct.c_void_p)))
    res = out.astype(np.float64)
    res[out == 255] = -1.0
    # chromosomes with fewer than two genes are skipped by .define_cnv_gene_regions (:1013): untouched
    chrs = np.asarray(infercnv_obj.gene_order.chr)
    for c in np.unique(chrs):
        rows = np.nonzero(chrs == c)[0]
        if rows.size < 2:
            res[rows] = np.asarray(infercnv_obj.expr_data)[rows]
    new = infercnv_obj.copy()
    new.expr_data = res
    return new


def _cell_groups(obj, by):
    """R/inferCNV_HMM.R:713-733 -> list of (name, 0-based index vector)."""
    if obj.tumor_subclusters is None:
        by = "consensus"
    if by == "consensus":
        d = dict(obj.reference_grouped_cell_indices)
        d.update(obj.observation_grouped_cell_indices)
        return [(k, np.asarray(v, dtype=np.int32)) for k, v in d.items()]
    if by == "subcluster":
        out = []
        for grp, subs in obj.tumor_subclusters["subclusters"].items():
            for name, v in subs.items():
                out.append((f"{grp}.{name}", np.asarray(v, dtype=np.int32)))   # unlist(recursive=FALSE) names
        return out
    if by == "cell":
        cells = obj.cells()
        order = np.concatenate([np.asarray(v, dtype=np.int32) for v in obj.reference_grouped_cell_indices.values()] +
                               [np.asarray(v, dtype=np.int32) for v in obj.observation_grouped_cell_indices.values()])
        return [(str(cells[i]), np.array([i], dtype=np.int32)) for i in order]
    raise ValueError("by must be one of consensus, subcluster, cell")


def get_predicted_CNV_regions(infercnv_obj: InfercnvObject, by="consensus"):
    """R/inferCNV_HMM.R:706-764.  Returns a list of dicts {cell_group_name, cells, gene_regions, cnv_ranges};
    gene_regions = ordered list of (region name, dict(state, gene idx array, chr, start, end arrays))."""
    groups = _cell_groups(infercnv_obj, by)
    cons = state_consensus(infercnv_obj, [g for _, g in groups])
    chrs = np.asarray(infercnv_obj.gene_order.chr)
    start = np.asarray(infercnv_obj.gene_order.start) if infercnv_obj.gene_order.start is not None else np.arange(chrs.size)
    stop = np.asarray(infercnv_obj.gene_order.stop) if infercnv_obj.gene_order.stop is not None else np.arange(chrs.size)
    _, first = np.unique(chrs, return_index=True)
    chr_order = chrs[np.sort(first)]
    cells = infercnv_obj.cells()
    out = []
    counter = 0
    for gi, (name, idx) in enumerate(groups):
        regions = []
        for c in chr_order:                                   # .define_cnv_gene_regions (:1005-1057)
            gene_idx = np.nonzero(chrs == c)[0]
            if gene_idx.size < 2:
                continue
            states = cons[gene_idx, gi]
            cuts = np.concatenate([[0], np.nonzero(states[1:] != states[:-1])[0] + 1, [gene_idx.size]])
            for a, b in zip(cuts[:-1], cuts[1:]):
                counter += 1
                rows = gene_idx[a:b]
                regions.append((f"{c}-region_{counter}",
                                {"state": float(states[a]), "gene": rows, "chr": c, "start": start[rows], "end": stop[rows]}))
        ranges = [(rn, r["state"], r["chr"], r["start"].min(), r["end"].max()) for rn, r in regions]   # :1071-1087
        out.append({"cell_group_name": name, "cells": cells[idx], "gene_regions": regions, "cnv_ranges": ranges})
    return out


def _fmt(v):
    """write.table(quote=FALSE) formatting of numbers: integers without a decimal point, up to 15 significant digits."""
    if isinstance(v, (float, np.floating)):
        return str(int(v)) if float(v).is_integer() else repr(float(np.float64(f"{v:.15g}")))
    return str(v)


def generate_cnv_region_reports(infercnv_obj: InfercnvObject, output_filename_prefix, out_dir, ignore_neutral_state=None,
                                by="consensus"):
    """R/inferCNV_HMM.R:790-869: writes <prefix>.cell_groupings, .pred_cnv_regions.dat, .pred_cnv_genes.dat and
    .genes_used.dat (tab separated, header row, no quotes, no row names except for genes_used)."""
    cnv_regions = get_predicted_CNV_regions(infercnv_obj, by)
    os.makedirs(out_dir, exist_ok=True)
    genes = infercnv_obj.genes()
    path = lambda suffix: os.path.join(out_dir, output_filename_prefix + suffix)
    with open(path(".cell_groupings"), "w") as fh:
        fh.write("cell_group_name\tcell\n")
        for x in cnv_regions:
            for c in x["cells"]:
                fh.write(f"{x['cell_group_name']}\t{c}\n")
    keep = (lambda s: True) if ignore_neutral_state is None else (lambda s: s != ignore_neutral_state)
    with open(path(".pred_cnv_regions.dat"), "w") as fh:
        fh.write("cell_group_name\tcnv_name\tstate\tchr\tstart\tend\n")
        for x in cnv_regions:
            for rn, state, c, s, e in x["cnv_ranges"]:
                if keep(state):
                    fh.write("\t".join([x["cell_group_name"], rn, _fmt(state), str(c), _fmt(s), _fmt(e)]) + "\n")
    with open(path(".pred_cnv_genes.dat"), "w") as fh:
        fh.write("cell_group_name\tgene_region_name\tstate\tgene\tchr\tstart\tend\n")
        for x in cnv_regions:
            for rn, r in x["gene_regions"]:
                if keep(r["state"]):
                    for g, s, e in zip(r["gene"], r["start"], r["end"]):
                        fh.write("\t".join([x["cell_group_name"], rn, _fmt(r["state"]), str(genes[g]), str(r["chr"]),
                                            _fmt(s), _fmt(e)]) + "\n")
    go = infercnv_obj.gene_order
    with open(path(".genes_used.dat"), "w") as fh:        # write.table(gene_order, quote=FALSE, sep="\t"): row names kept
        fh.write("chr\tstart\tstop\n")
        n = np.asarray(go.chr).size
        st = go.start if go.start is not None else np.arange(n)
        sp = go.stop if go.stop is not None else np.arange(n)
        for i in range(n):
            fh.write(f"{genes[i]}\t{np.asarray(go.chr)[i]}\t{_fmt(st[i])}\t{_fmt(sp[i])}\n")
    return cnv_regions
