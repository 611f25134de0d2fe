"""Python mirror of the reference's S4 class `infercnv` (R/inferCNV.R:37-47).

The class is the data model of the drop-in boundary: every step function takes
an object and returns an object, touching only `expr_data` (and, by recursion,
`hspike`), exactly like the R functions called from run()
(R/inferCNV_ops.R:771-1589).  Index vectors are 0-based here (R: 1-based).
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np


@dataclass
class GeneOrder:
    """gene_order data.frame(chr, start, stop); `chr` kept as strings/codes."""
    chr: np.ndarray
    start: Optional[np.ndarray] = None
    stop: Optional[np.ndarray] = None


@dataclass
class InfercnvObject:
    expr_data: np.ndarray                                   # genes x cells, float64
    gene_order: GeneOrder
    reference_grouped_cell_indices: Dict[str, np.ndarray] = field(default_factory=dict)
    observation_grouped_cell_indices: Dict[str, np.ndarray] = field(default_factory=dict)
    count_data: Optional[np.ndarray] = None
    tumor_subclusters: Optional[dict] = None                # {"subclusters": {group: {name: idx}}}
    options: dict = field(default_factory=dict)
    hspike: Optional["InfercnvObject"] = None               # R slot `.hspike`
    gene_names: Optional[np.ndarray] = None                 # rownames(expr.data)
    cell_names: Optional[np.ndarray] = None                 # colnames(expr.data)

    def genes(self):
        return self.gene_names if self.gene_names is not None else np.array(
            [f"gene_{i + 1}" for i in range(self.expr_data.shape[0])])

    def cells(self):
        return self.cell_names if self.cell_names is not None else np.array(
            [f"cell_{i + 1}" for i in range(self.expr_data.shape[1])])

    # ---- helpers mirroring R/inferCNV.R ------------------------------------
    def has_reference_cells(self) -> bool:
        """has_reference_cells (R/inferCNV.R:510-514)."""
        return len(self.reference_grouped_cell_indices) > 0

    def get_reference_grouped_cell_indices(self) -> np.ndarray:
        """unlist(reference_grouped_cell_indices) (R/inferCNV.R:517-520)."""
        if not self.reference_grouped_cell_indices:
            return np.zeros(0, dtype=np.int32)
        return np.concatenate([np.asarray(v, dtype=np.int32) for v in self.reference_grouped_cell_indices.values()])

    def ref_groups_or_proxy(self) -> List[np.ndarray]:
        """Reference groups, or one 'proxyNormal' group of all observation cells
        when no references exist (R/inferCNV_ops.R:1683-1688)."""
        if self.has_reference_cells():
            return [np.asarray(v, dtype=np.int32) for v in self.reference_grouped_cell_indices.values()]
        return [np.concatenate([np.asarray(v, dtype=np.int32)
                                for v in self.observation_grouped_cell_indices.values()])]

    def chr_layout(self):
        """(perm, chr_start): `perm` gathers genes into per-chromosome contiguous
        blocks in order of first appearance (unique(gene_order$chr)); identity when
        the object is already ordered (.order_reduce, R/inferCNV.R:352-428)."""
        chrs = np.asarray(self.gene_order.chr)
        _, first, inv = np.unique(chrs, return_index=True, return_inverse=True)
        rank_of_code = np.argsort(np.argsort(first))          # code -> order of first appearance
        key = rank_of_code[inv]
        perm = np.argsort(key, kind="stable")
        sizes = np.bincount(key, minlength=first.size)
        chr_start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        identity = bool(np.all(perm == np.arange(perm.size)))
        return (None if identity else perm), chr_start

    def copy(self) -> "InfercnvObject":
        return copy.copy(self)

    def validate(self):
        """validate_infercnv_obj (R/inferCNV.R:471-505), the checks relevant here."""
        G, C = self.expr_data.shape
        if np.asarray(self.gene_order.chr).shape[0] != G:
            raise ValueError("gene_order rows must match expr_data rows")
        for name, idx in list(self.reference_grouped_cell_indices.items()) + \
                list(self.observation_grouped_cell_indices.items()):
            idx = np.asarray(idx)
            if idx.size and (idx.min() < 0 or idx.max() >= C):
                raise ValueError(f"cell indices of group {name!r} out of range")
        return True
