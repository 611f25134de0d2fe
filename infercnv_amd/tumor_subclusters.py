"""Host-side mirror of the distance step of the reference's subclustering
(R/inferCNV_tumor_subclusters.R:180-194: `hclust(parallelDist(t(tumor_expr_data)))`): the Euclidean distances
between the cells of one tumor group, computed on the GPU (icnv_cell_distances_dev, fp64 matrix cores).  The
clustering itself (hclust / Leiden) stays in R -- SURVEY.md 8f #4 scopes only the dense contraction."""
from __future__ import annotations

import numpy as np

from . import device
from .infercnv_object import InfercnvObject


def parallelDist(infercnv_obj: InfercnvObject, cells, as_dist: bool = True):
    """parallelDist(t(expr.data[, cells]), method="euclidean").

    as_dist=True returns the R `dist` object's vector (lower triangle in column order == SciPy's condensed
    form); False the full symmetric (n, n) matrix."""
    import torch
    cells = np.asarray(cells, dtype=np.int32)
    if cells.ndim != 1 or cells.size < 1:
        raise ValueError("cells must be a non-empty index vector")
    x = infercnv_obj.expr_data
    xd = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float64).T)).cuda()
    d = device.cell_distances(xd, cells).cpu().numpy()
    if not as_dist:
        return d
    iu = np.triu_indices(cells.size, k=1)
    return d[iu]
