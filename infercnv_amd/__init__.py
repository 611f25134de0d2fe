"""infercnv_amd -- MI355X-native hot path of inferCNV (smoothing chain, i6/i3
HMM Viterbi, 2-D median denoise) behind the reference's step-function API.

Host mirror of the R functions: `ops`, `hmm`, `noise_reduction`
(InfercnvObject in, InfercnvObject out).  Device-resident API: `device`.
Cell-sharded multi-GPU: `sharded`.  All compute is in libicnv_hip.so
(hand-written HIP for gfx950); there is no CPU fallback.
"""
from ._lib import IcnvError, LIB_PATH, load  # noqa: F401
from .infercnv_object import GeneOrder, InfercnvObject  # noqa: F401

__all__ = ["IcnvError", "LIB_PATH", "load", "GeneOrder", "InfercnvObject"]
