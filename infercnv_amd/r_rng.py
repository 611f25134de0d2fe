"""R's default random number stream, restated: `set.seed(seed)` + `sample(x, size, replace = TRUE)` as base R >= 3.6
produces them (RNGkind "Mersenne-Twister", "Inversion", sample.kind "Rejection").

Why it is here: `get_hspike_cnv_mean_sd_trend_by_num_cells_fit` (R/inferCNV_HMM.R:154-212) resamples the hidden spike-in's
residuals with `sample()`; its result is a function of R's RNG state.  The reference never seeds that state (no set.seed in
/root/reference/R), so its own output differs from session to session; what CAN be matched is: same seed -> same stream ->
same fit.  Base R is a dependency outside /root/reference (DESCRIPTION: R >= 4.0); the algorithm below follows its
published sources:

  * src/main/RNG.c, RNG_Init: the seed is scrambled by 50 steps of the LCG s <- 69069 s + 1 (mod 2^32), then the 625
    words of the Mersenne-Twister's seed vector are the next 625 LCG values; FixupSeeds sets the first word (mti) to 624,
    so the first draw regenerates the state.  The generator itself is MT19937 (Matsumoto & Nishimura 1998,
    genrand_int32) -- NumPy's MT19937 bit generator runs the same recurrence, so the state words are handed to it.
  * unif_rand() = fixup(genrand_int32 * 2^-32) (fixup moves exact 0 / 1 inside the open interval).
  * src/main/RNG.c, R_unif_index(dn) with rejection sampling: bits = ceil(log2(dn)); rbits(bits) concatenates
    floor(unif_rand() * 65536) -- the top 16 bits of a draw -- for n = 0, 16, ... <= bits, masks the low `bits` bits;
    values >= dn are rejected.  Every attempt consumes the same number of draws, so a whole batch of attempts is formed
    at once and the accepted ones, in order, are R's stream.
  * src/main/random.c, do_sample with replace = TRUE: index i is R_unif_index(n) + 1.

Pinned by the outputs every R >= 3.6 user knows (tests/test_oracle.py): set.seed(42); runif(3) = 0.9148060 0.9370754
0.2861395; set.seed(123); sample(1:100, 5) = 31 79 51 14 67; set.seed(42); sample(1:10) = 1 5 10 8 2 4 6 9 7 3.
"""
from __future__ import annotations

import math

import numpy as np


class RRandom:
    """R's default generator after `set.seed(seed)`."""

    def __init__(self, seed: int):
        s = np.uint32(int(seed) & 0xFFFFFFFF)
        with np.errstate(over="ignore"):
            for _ in range(50):                               # initial scrambling
                s = np.uint32(np.uint32(69069) * s + np.uint32(1))
            words = np.empty(625, dtype=np.uint32)
            for j in range(625):
                s = np.uint32(np.uint32(69069) * s + np.uint32(1))
                words[j] = s
        self._bg = np.random.MT19937()
        self._bg.state = {"bit_generator": "MT19937", "state": {"key": words[1:].copy(), "pos": 624}}   # words[0] is mti <- 624

    # ---- draws
    def raw(self, n: int) -> np.ndarray:
        """The next n outputs of genrand_int32 (uint32 values in a uint64 array)."""
        return self._bg.random_raw(int(n))

    def unif_rand(self, n: int) -> np.ndarray:
        u = self.raw(n).astype(np.float64) * 2.3283064365386963e-10
        i2_32m1 = 2.328306437080797e-10
        u[u <= 0.0] = 0.5 * i2_32m1
        u[(1.0 - u) <= 0.0] = 1.0 - 0.5 * i2_32m1
        return u

    def unif_index(self, dn: int, size: int) -> np.ndarray:
        """`size` values of R_unif_index(dn) (0-based), rejection sampling."""
        dn = int(dn)
        if dn <= 0:
            return np.zeros(size, dtype=np.int64)
        if dn == 1:
            bits = 0
        else:
            bits = int(math.ceil(math.log2(dn)))
        k = bits // 16 + 1                                    # draws per attempt: n = 0, 16, ... <= bits
        if k > 3:
            raise ValueError("population too large for this restatement (> 2^47 elements)")
        mask = (1 << bits) - 1
        out = np.empty(size, dtype=np.int64)
        got = 0
        while got < size:
            need = size - got
            att = max(16, int(need * 2.2) + 16)               # at most half of the attempts are rejected
            st = self._bg.state                               # attempts beyond the last accepted one must not be consumed
            r = self.raw(att * k).reshape(att, k) >> np.uint64(16)
            v = np.zeros(att, dtype=np.uint64)
            for j in range(k):
                v = v * np.uint64(65536) + r[:, j]
            v = (v & np.uint64(mask)).astype(np.int64)
            ok = np.nonzero(v < dn)[0]
            take = ok[:need]
            out[got:got + take.size] = v[take]
            got += take.size
            used_attempts = (int(take[-1]) + 1) if (take.size == need and take.size) else att
            if used_attempts < att:                           # rewind to just behind the last attempt that was needed
                self._bg.state = st
                self.raw(used_attempts * k)
        return out

    def sample_replace(self, n: int, size: int) -> np.ndarray:
        """sample.int(n, size, replace = TRUE) - 1 (0-based indices)."""
        return self.unif_index(n, size)

    def sample_perm(self, n: int) -> np.ndarray:
        """sample.int(n) - 1: the permutation of do_sample without replacement (x[j] <- x[--n] after each pick)."""
        x = np.arange(n, dtype=np.int64)
        out = np.empty(n, dtype=np.int64)
        m = n
        for i in range(n):
            j = int(self.unif_index(m, 1)[0])
            out[i] = x[j]
            m -= 1
            x[j] = x[m]
        return out


# ---------------------------------------------------------------------------------------------------------------------
# rnorm() and qnorm(): what get_HoneyBADGER_setGexpDev (R/inferCNV_i3HMM.R:469-493) draws its samples with.
#   * src/nmath/snorm.c, norm_rand() with N01_kind = INVERSION (R's default): BIG = 2^27;
#       u = unif_rand(); u = (int)(BIG * u) + unif_rand(); return qnorm5(u / BIG, 0, 1, TRUE, FALSE)
#     (two uniforms per normal deviate: the first supplies the high 27 bits).
#   * src/nmath/qnorm.c: Wichura's algorithm AS 241 (Appl. Statist. 37, 1988), routine PPND16 -- the published
#     coefficients below; the branch R >= 4.3 adds for |log p| beyond r = 27 cannot be reached from u / BIG.
_AS241_A = (3.3871328727963666080e0, 1.3314166789178437745e+2, 1.9715909503065514427e+3, 1.3731693765509461125e+4,
            4.5921953931549871457e+4, 6.7265770927008700853e+4, 3.3430575583588128105e+4, 2.5090809287301226727e+3)
_AS241_B = (1.0, 4.2313330701600911252e+1, 6.8718700749205790830e+2, 5.3941960214247511077e+3, 2.1213794301586595867e+4,
            3.9307895800092710610e+4, 2.8729085735721942674e+4, 5.2264952788528545610e+3)
_AS241_C = (1.42343711074968357734e0, 4.63033784615654529590e0, 5.76949722146069140550e0, 3.64784832476320460504e0,
            1.27045825245236838258e0, 2.41780725177450611770e-1, 2.27238449892691845833e-2, 7.74545014278341407640e-4)
_AS241_D = (1.0, 2.05319162663775882187e0, 1.67638483018380384940e0, 6.89767334985100004550e-1, 1.48103976427480074590e-1,
            1.51986665636164571966e-2, 5.47593808499534494600e-4, 1.05075007164441684324e-9)
_AS241_E = (6.65790464350110377720e0, 5.46378491116411436990e0, 1.78482653991729133580e0, 2.96560571828504891230e-1,
            2.65321895265761230930e-2, 1.24266094738807843860e-3, 2.71155556874348757815e-5, 2.01033439929228813265e-7)
_AS241_F = (1.0, 5.99832206555887937690e-1, 1.36929880922735805310e-1, 1.48753612908506148525e-2, 7.86869131145613259100e-4,
            1.84631831751005468180e-5, 1.42151175831644588870e-7, 2.04426310338993978564e-15)


def _horner(c, r):
    p = np.full_like(r, c[-1])
    for v in c[-2::-1]:
        p = p * r + v
    return p


def qnorm(p):
    """qnorm(p) (lower tail, mean 0, sd 1) by AS 241 / PPND16, vectorised; p in (0, 1)."""
    p = np.asarray(p, dtype=np.float64)
    q = p - 0.5
    out = np.empty_like(p)
    mid = np.abs(q) <= 0.425
    r = 0.180625 - q[mid] * q[mid]
    out[mid] = q[mid] * _horner(_AS241_A, r) / _horner(_AS241_B, r)
    rest = ~mid
    if rest.any():
        qq = q[rest]
        r = np.sqrt(-np.log(np.where(qq < 0, p[rest], 1.0 - p[rest])))
        val = np.empty_like(r)
        lo = r <= 5.0
        rl = r[lo] - 1.6
        val[lo] = _horner(_AS241_C, rl) / _horner(_AS241_D, rl)
        rh = r[~lo] - 5.0
        val[~lo] = _horner(_AS241_E, rh) / _horner(_AS241_F, rh)
        out[rest] = np.where(qq < 0, -val, val)
    return out


def _norm_rand(self, n: int) -> np.ndarray:
    """n draws of norm_rand() (INVERSION)."""
    u = self.unif_rand(2 * int(n)).reshape(int(n), 2)
    big = 134217728.0
    return qnorm((np.floor(big * u[:, 0]) + u[:, 1]) / big)


def _rnorm(self, n: int, mean: float = 0.0, sd: float = 1.0) -> np.ndarray:
    """rnorm(n, mean, sd) = mean + sd * norm_rand()  (src/nmath/rnorm.c)."""
    return mean + sd * self.norm_rand(n)


RRandom.norm_rand = _norm_rand
RRandom.rnorm = _rnorm
