#!/usr/bin/env python3
"""Generate median9x9_net.h: straight-line min/max networks for the exact median of a full
9 x 9 window whose nine columns are already sorted (the interior fast path of the 2-D median
denoise, R/noise_reduction.R:92-113 with window_size = 7 -> (7+2)^2 = 81 values).

  python infercnv_amd/csrc/gen_median_net.py        # rewrites infercnv_amd/csrc/median9x9_net.h

Scheme (Batcher odd-even merges, pruned by liveness):
  sort9            : 25 compare-exchanges (used for the shared column sorts)
  L1: merge 9+9   -> 18   (four times: columns 0|1, 2|3, 4|5, 6|7)
  L2: merge 18+18 -> 36   (twice)
  L3: merge 36+36 -> only merged positions 31..40 are kept: an element at position i of the 72 has
      overall rank in [i, i+9], so only 31 <= i <= 40 can be the median (rank 40 of 81)
  L4: merge those 10 with column 8 (9 values) -> position 9 of the 19 is the median.
The generator tracks +inf padding symbolically (no code for compare-exchanges with padding),
renames instead of moving, and drops every min/max whose result is dead.  It verifies the sort9
network exhaustively (0-1 principle) and the whole procedure on random inputs with ties.
"""
import os
import random
import sys

SORT9 = [(0, 3), (1, 7), (2, 5), (4, 8), (0, 7), (2, 4), (3, 8), (5, 6), (0, 2), (1, 3), (4, 5), (7, 8),
         (1, 4), (3, 6), (5, 7), (0, 1), (2, 4), (3, 5), (6, 8), (2, 3), (4, 5), (6, 7), (1, 2), (3, 4), (5, 6)]


def oddeven_merge(lo, n, r):
    """Batcher's odd-even merge of the two sorted halves of [lo, lo+n) (n a power of two), stride r."""
    step = r * 2
    if step < n:
        yield from oddeven_merge(lo, n, step)
        yield from oddeven_merge(lo + r, n, step)
        for i in range(lo + r, lo + n - r, step):
            yield (i, i + r)
    else:
        yield (lo, lo + r)


class Net:
    """Symbolic builder: values are SSA names, 'INF' is padding; records (dst, op, a, b)."""

    def __init__(self):
        self.ops = []
        self.n = 0

    def new(self):
        self.n += 1
        return f"t{self.n}"

    def ce(self, vals, i, j):
        a, b = vals[i], vals[j]
        if b == "INF":
            return                      # min stays, max stays +inf
        if a == "INF":
            vals[i], vals[j] = b, a     # pure renaming
            return
        lo, hi = self.new(), self.new()
        self.ops.append((lo, "ICNV_FMIN", a, b))
        self.ops.append((hi, "ICNV_FMAX", a, b))
        vals[i], vals[j] = lo, hi

    def merge(self, A, B):
        """Merge two sorted symbolic lists -> sorted list (len(A)+len(B))."""
        n = 1
        while n < max(len(A), len(B)):
            n *= 2
        vals = A + ["INF"] * (n - len(A)) + B + ["INF"] * (n - len(B))
        for (i, j) in oddeven_merge(0, 2 * n, 1):
            self.ce(vals, i, j)
        out = vals[:len(A) + len(B)]
        assert all(v != "INF" for v in out) and all(v == "INF" for v in vals[len(A) + len(B):])
        return out

    def prune(self, live):
        live = set(live)
        kept = []
        for dst, op, a, b in reversed(self.ops):
            if dst in live:
                kept.append((dst, op, a, b))
                live.add(a)
                live.add(b)
        self.ops = kept[::-1]


def build_median81(pair=False):
    net = Net()
    cols = [[f"a[{9 * c + k}]" for k in range(9)] for c in range(9)]
    m01, m23 = net.merge(cols[0], cols[1]), net.merge(cols[2], cols[3])
    m45, m67 = net.merge(cols[4], cols[5]), net.merge(cols[6], cols[7])
    A, B = net.merge(m01, m23), net.merge(m45, m67)
    AB = net.merge(A, B)
    if pair:     # ranks 40 and 41: positions 31..41 of the 72 can hold them
        fin = net.merge(AB[31:42], cols[8])
        result = (fin[9], fin[10])
        net.prune(list(result))
    else:
        fin = net.merge(AB[31:41], cols[8])
        result = fin[9]
        net.prune([result])
    return net, result


def build_shared_window():
    """Two outputs that are neighbours along the column axis share eight of their nine sorted columns: positions
    31..40 of the merged 72 shared values (the only ones that can be the median of 72 + 9) are computed once."""
    net = Net()
    cols = [[f"s[{9 * c + k}]" for k in range(9)] for c in range(8)]
    m01, m23 = net.merge(cols[0], cols[1]), net.merge(cols[2], cols[3])
    m45, m67 = net.merge(cols[4], cols[5]), net.merge(cols[6], cols[7])
    AB = net.merge(net.merge(m01, m23), net.merge(m45, m67))
    win = AB[31:41]
    net.prune(win)
    return net, win


def build_window_finish():
    """Median of the 81 = position 9 of the merge of the shared window (10 sorted values) with the output's own
    sorted column (9 values)."""
    net = Net()
    fin = net.merge([f"w[{k}]" for k in range(10)], [f"p[{k}]" for k in range(9)])
    net.prune([fin[9]])
    return net, fin[9]


def evaluate_env(net, env):
    for dst, op, x, y in net.ops:
        env[dst] = min(env[x], env[y]) if op == "ICNV_FMIN" else max(env[x], env[y])
    return env


def evaluate(net, result, a):
    env = {f"a[{i}]": a[i] for i in range(81)}
    for dst, op, x, y in net.ops:
        env[dst] = min(env[x], env[y]) if op == "ICNV_FMIN" else max(env[x], env[y])
    return tuple(env[r] for r in result) if isinstance(result, tuple) else env[result]


def main():
    # sort9: exhaustive 0-1 check
    for bits in range(512):
        v = [(bits >> i) & 1 for i in range(9)]
        for i, j in SORT9:
            if v[i] > v[j]:
                v[i], v[j] = v[j], v[i]
        assert v == sorted(v), "sort9 network is wrong"
    net, result = build_median81()
    net2, result2 = build_median81(pair=True)
    rng = random.Random(7)
    for trial in range(3000):
        mode = trial % 4
        if mode == 0:
            vals = [rng.gauss(0, 1) for _ in range(81)]
        elif mode == 1:
            vals = [float(rng.randint(0, 5)) for _ in range(81)]           # heavy ties
        elif mode == 2:
            vals = [0.0] * 81
            for _ in range(rng.randint(0, 81)):
                vals[rng.randrange(81)] = rng.gauss(0, 1)
        else:
            vals = [float(i) for i in range(81)]
            rng.shuffle(vals)
        a = []
        for c in range(9):
            a += sorted(vals[9 * c:9 * c + 9])
        assert evaluate(net, result, a) == sorted(vals)[40], "median network is wrong"
        assert evaluate(net2, result2, a) == (sorted(vals)[40], sorted(vals)[41]), "pair network is wrong"
        # clamped windows: m real values padded with -inf / +inf so that the wanted ranks land on 40 (and 41)
        m = rng.choice([25, 30, 35, 36, 40, 42, 45, 48, 49, 54, 56, 63, 64, 72])
        real = vals[:m]
        n_lo = (81 - m) // 2 if m % 2 else 41 - m // 2
        padded = real + [float("-inf")] * n_lo + [float("inf")] * (81 - m - n_lo)
        rng.shuffle(padded)
        ap = []
        for c in range(9):
            ap += sorted(padded[9 * c:9 * c + 9])
        r40, r41 = evaluate(net2, result2, ap)
        sr = sorted(real)
        want = sr[m // 2] if m % 2 else (sr[m // 2 - 1] + sr[m // 2]) * 0.5
        got = r40 if m % 2 else (r40 + r41) * 0.5
        assert got == want, "padded selection is wrong"
    # shared-window pair: ten sorted columns, outputs over columns 0..8 and 1..9
    netw, win = build_shared_window()
    netf, resf = build_window_finish()
    for trial in range(2000):
        mode = trial % 3
        if mode == 0:
            vals = [rng.gauss(0, 1) for _ in range(90)]
        elif mode == 1:
            vals = [float(rng.randint(0, 4)) for _ in range(90)]
        else:
            vals = [float(i) for i in range(90)]
            rng.shuffle(vals)
        cols = [sorted(vals[9 * c:9 * c + 9]) for c in range(10)]
        env = {f"s[{9 * c + k}]": cols[c + 1][k] for c in range(8) for k in range(9)}
        env = evaluate_env(netw, env)
        w = [env[x] for x in win]
        assert w == sorted(w)
        for own, lo in ((cols[0], 0), (cols[9], 9)):
            e2 = {f"w[{k}]": w[k] for k in range(10)}
            e2.update({f"p[{k}]": own[k] for k in range(9)})
            e2 = evaluate_env(netf, e2)
            want = sorted(vals[lo:lo + 81])[40]
            assert e2[resf] == want, "shared-window pair network is wrong"
    n_ops = len(net.ops)
    lines = [
        "// GENERATED by gen_median_net.py -- do not edit.",
        f"// Exact median of 81 values given as nine sorted columns a[9*c + k] (k ascending): {n_ops} min/max",
        "// operations (Batcher odd-even merges 9+9, 18+18, a pruned 36+36 and a pruned 10+9).",
        "#pragma once",
        "#ifndef ICNV_FMIN   // the including file may map these to the bare v_min_f64 / v_max_f64",
        "#define ICNV_FMIN(a, b) fmin(a, b)",
        "#define ICNV_FMAX(a, b) fmax(a, b)",
        "#endif",
        "#define ICNV_SORT9(v) do { \\",
    ]
    for i, j in SORT9:
        lines.append(f"    {{ const double lo_ = ICNV_FMIN(v[{i}], v[{j}]); v[{j}] = ICNV_FMAX(v[{i}], v[{j}]); v[{i}] = lo_; }} \\")
    lines.append("} while (0)")
    lines.append("")
    lines.append("__device__ inline double median81_sorted_columns(const double (&a)[81]) {")
    for dst, op, x, y in net.ops:
        lines.append(f"    const double {dst} = {op}({x}, {y});")
    lines.append(f"    return {result};")
    lines.append("}")
    lines.append("")
    lines.append(f"// ranks 40 and 41 of the 81 values ({len(net2.ops)} min/max): clamped (border) windows are padded with")
    lines.append("// -inf / +inf so that the wanted order statistics of the real values land on these ranks")
    lines.append("__device__ inline void median81_pair_sorted_columns(const double (&a)[81], double &r40, double &r41) {")
    for dst, op, x, y in net2.ops:
        lines.append(f"    const double {dst} = {op}({x}, {y});")
    lines.append(f"    r40 = {result2[0]};")
    lines.append(f"    r41 = {result2[1]};")
    lines.append("}")
    lines.append("")
    lines.append(f"// Two outputs that are neighbours along the column axis share eight sorted columns s[9*c + k]: positions 31..40 of")
    lines.append(f"// their merged 72 values ({len(netw.ops)} min/max, once per pair) ...")
    lines.append("__device__ inline void median72_window(const double (&s)[72], double (&w)[10]) {")
    for dst, op, x, y in netw.ops:
        lines.append(f"    const double {dst} = {op}({x}, {y});")
    for k, x in enumerate(win):
        lines.append(f"    w[{k}] = {x};")
    lines.append("}")
    lines.append("")
    lines.append(f"// ... and each output's median is position 9 of that window merged with its own sorted column ({len(netf.ops)} min/max)")
    lines.append("__device__ inline double median_window_finish(const double (&w)[10], const double (&p)[9]) {")
    for dst, op, x, y in netf.ops:
        lines.append(f"    const double {dst} = {op}({x}, {y});")
    lines.append(f"    return {resf};")
    lines.append("}")
    here = os.path.dirname(os.path.abspath(__file__))
    text = "\n".join(lines) + "\n"
    if "--check" in sys.argv:      # tests: the committed header is what this generator (and its verification) produces
        if open(os.path.join(here, "median9x9_net.h")).read() != text:
            raise SystemExit("median9x9_net.h is stale: run gen_median_net.py")
        print("median9x9_net.h is up to date")
        return
    with open(os.path.join(here, "median9x9_net.h"), "w") as fh:
        fh.write(text)
    print(f"median81 network: {n_ops} min/max ops; sort9: {len(SORT9)} compare-exchanges; "
          f"shared window: {len(netw.ops)} per pair + {len(netf.ops)} per output")


if __name__ == "__main__":
    main()
