#!/usr/bin/env python3
"""Developer tool: ablation builds of the chain2 kernel (no GPU needed to build).  Each variant is the product source with
a few textual patches (a piece left out), compiled as a replacement translation unit into exp_libs/lib_c2_<name>.so by
scripts/build_variant.sh; scripts/bench_libs.sh then times them side by side on one box (LIBS="c2_base c2_nostore ...").
The results of a left-out piece are wrong by construction: only the launch times are compared."""
import os, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = open(os.path.join(root, "infercnv_amd", "csrc", "chain2.hip")).read()
V = {
    "base": [],
    "nostore": [("            if (slot_valid(sl)) {\n                if (dpre) store_vec_stream", "            if (slot_valid(sl) && a.n_cells < 0) {\n                if (dpre) store_vec_stream")],
    "onestore": [("        char *dpre = (MODE == MODE_APPLY && a.pre_out) ?", "        char *dpre = (MODE == MODE_APPLY && a.pre_out && a.n_cells < 0) ?")],
    "nomedian": [("            bool guessed = range_known;\n            double lo, hi;\n          measure_range:", "            bool guessed = range_known;\n            double lo, hi;\n            if (a.n_cells > 0) goto median_done;\n          measure_range:"),
                 ("        // ---------------- steps 11 (subtract), 12, 14, 22 and the stores", "      median_done:\n        // ---------------- steps 11 (subtract), 12, 14, 22 and the stores")],
    "nobounds": [("load_vec<2>(reinterpret_cast<const double *>(b1lo + go[s]), lo1[j]);", "lo1[j][0] = lo1[j][1] = -1.0;"),
                 ("load_vec<2>(reinterpret_cast<const double *>(b1hi + go[s]), hi1[j]);", "hi1[j][0] = hi1[j][1] = 1.0;"),
                 ("load_vec<2>(reinterpret_cast<const double *>(b2lo + go0[s]), lo2[s]);", "lo2[s][0] = lo2[s][1] = -1.0;"),
                 ("load_vec<2>(reinterpret_cast<const double *>(b2hi + go0[s]), hi2[s]);", "hi2[s][0] = hi2[s][1] = 1.0;"),
                 ("load_vec<2>(reinterpret_cast<const double *>(b2lo + go1[s]), lo2[s]);", "lo2[s][0] = lo2[s][1] = -1.0;"),
                 ("load_vec<2>(reinterpret_cast<const double *>(b2hi + go1[s]), hi2[s]);", "hi2[s][0] = hi2[s][1] = 1.0;")],
    "noexp2": [("                    y[0] = exp2_lean(y[0]);\n                    y[1] = exp2_lean(y[1]);", "                    y[0] = y[0] + 1.0;\n                    y[1] = y[1] + 1.0;")],
}
want = sys.argv[1:] or list(V)
for name in want:
    s = src
    for old, new in V[name]:
        assert s.count(old) >= 1, (name, old[:60])
        s = s.replace(old, new)
    d = f"/tmp/c2var/{name}"
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "chain2.hip"), "w").write(s)
    subprocess.run(["bash", os.path.join(root, "scripts", "build_variant.sh"), "c2_" + name, "", os.path.join(d, "chain2.hip")],
                   check=True, env=dict(os.environ, REBUILD=" "))
