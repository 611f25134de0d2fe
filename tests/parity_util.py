"""Shared assertions of the parity tests.

Step 22 (`clear_noise_via_ref_mean_sd`, R/inferCNV_ops.R:2335) is a STRICT select: `x > mu - s & x < mu + s -> mu`.
Two correct fp64 implementations whose pre-denoise values agree to `tol` can therefore disagree on an element only
when that element lies within `tol` of one of the two bounds -- and then one of them returns `mu`, the other the
pre-denoise value.  `check_denoise_flips` accepts exactly that and nothing else: every differing element is tied to a
bound, and the number of such elements is counted, printed and (where the caller knows it) asserted.
"""
import numpy as np


def check_denoise_flips(got_out, ref_out, ref_pre, mu, s, tol=1e-11, expect=None, label=""):
    """got_out / ref_out: denoised matrices (HIP path / oracle); ref_pre: the oracle's matrix before step 22;
    (mu, s): the oracle's step-22 parameters.  Returns the number of elements on which the select differs."""
    got_out = np.asarray(got_out, dtype=np.float64)
    ref_out = np.asarray(ref_out, dtype=np.float64)
    ref_pre = np.asarray(ref_pre, dtype=np.float64)
    scale = max(1.0, float(np.abs(ref_pre).max())) if ref_pre.size else 1.0
    atol = tol * scale
    diff = np.abs(got_out - ref_out) > atol
    n = int(diff.sum())
    if n:
        lo, hi = mu - s, mu + s
        p = ref_pre[diff]
        dist = np.minimum(np.abs(p - lo), np.abs(p - hi))
        # the pre-denoise value of a flipped element is within the chain tolerance of a bound ...
        assert (dist <= 2.0 * atol).all(), (label, n, float(dist.max()))
        # ... and the HIP path returned one of the two legitimate values for it
        g = got_out[diff]
        assert ((np.abs(g - mu) <= atol) | (np.abs(g - p) <= atol)).all(), (label, n)
    print(f"[denoise flips] {label or 'chain'}: {n} of {got_out.size} elements differ in the strict select")
    if expect is not None:
        assert n == expect, (label, n, expect)
    return n


def check_denoise_flips_t(got_out, ref_out, ref_pre, mu, s, tol=1e-11, expect=None, label=""):
    """`check_denoise_flips` on torch tensors (CUDA or CPU) of any shape: the same rule, evaluated where the tensors
    live -- the full-size tests compare 5e8 elements per matrix and do it on the GPU."""
    import torch
    scale = max(1.0, float(ref_pre.abs().max())) if ref_pre.numel() else 1.0
    atol = tol * scale
    diff = (got_out - ref_out).abs() > atol
    n = int(diff.sum())
    if n:
        lo, hi = mu - s, mu + s
        p = ref_pre[diff]
        dist = torch.minimum((p - lo).abs(), (p - hi).abs())
        assert bool((dist <= 2.0 * atol).all()), (label, n, float(dist.max()))
        g = got_out[diff]
        assert bool((((g - mu).abs() <= atol) | ((g - p).abs() <= atol)).all()), (label, n)
    print(f"[denoise flips] {label or 'chain'}: {n} of {got_out.numel()} elements differ in the strict select")
    if expect is not None:
        assert n == expect, (label, n, expect)
    return n
