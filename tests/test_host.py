"""CPU tests of the host side: the C-ABI library loads and exports every symbol
include/icnv.h declares (no compute without a GPU), argument validation, the
S4-mirror's layout logic, the synthetic generator, and the cell-sharding logic
of the multi-GPU path (world_size 2, gloo) with an oracle-backed engine.
"""
import ctypes as ct
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import infercnv_amd
    from infercnv_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = infercnv_amd.load()
    header = open(os.path.join(ROOT, "include", "icnv.h")).read()
    declared = set(re.findall(r"\b(icnv_[a-z0-9_]+)\s*\(", header))
    declared -= {"icnv_chain_cfg", "icnv_chain_t"}
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/icnv.h but not exported"
    assert declared <= set(_lib.PROTOTYPES), declared - set(_lib.PROTOTYPES)
    assert L.icnv_version() >= 100


def test_no_gpu_fails_loudly_not_silently():
    """Without a GPU the product path must raise -- there is no CPU fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from infercnv_amd import GeneOrder, IcnvError, InfercnvObject, ops
    obj = InfercnvObject(expr_data=np.zeros((4, 3)), gene_order=GeneOrder(chr=np.array(["a"] * 4)),
                         reference_grouped_cell_indices={"n": np.array([0])},
                         observation_grouped_cell_indices={"t": np.array([1, 2])})
    with pytest.raises(IcnvError):
        ops.subtract_ref_expr_from_obs(obj)


def test_argument_validation_happens_before_any_device_work():
    from infercnv_amd import _lib
    L = _lib.load()
    h = ct.c_void_p()
    bad_chr = _lib.Cfg(10, 4, [0, 5, 9], [[0]])            # chr_start does not end at G
    assert L.icnv_chain_begin(ct.byref(h), bad_chr.ptr()) == _lib.ERR_ARG
    assert b"chr_start" in L.icnv_last_error()
    even = _lib.Cfg(10, 4, [0, 10], [[0]], window_length=100)
    assert L.icnv_chain_begin(ct.byref(h), even.ptr()) == _lib.ERR_ARG
    oob = _lib.Cfg(10, 4, [0, 10], [[7]])                  # reference index outside the matrix
    assert L.icnv_chain_begin(ct.byref(h), oob.ptr()) == _lib.ERR_ARG
    big = _lib.Cfg(30000, 4, [0, 30000], [[0]])
    assert L.icnv_chain_begin(ct.byref(h), big.ptr()) == _lib.ERR_UNSUPPORTED     # one chromosome beyond the LDS
    assert b"chromosome" in L.icnv_last_error()
    wide = _lib.Cfg(30000, 4, [0, 14000, 30000], [[0]])    # beyond the fused kernel: the three-pass chain takes it
    assert L.icnv_chain_begin(ct.byref(h), wide.ptr()) == _lib.OK
    L.icnv_chain_end(h)
    ok = _lib.Cfg(10, 4, [0, 4, 10], [[0], [3, 1]])
    assert L.icnv_chain_begin(ct.byref(h), ok.ptr()) == _lib.OK
    assert L.icnv_chain_num_rounds(h) == 3                 # steps 8, 12, 22
    L.icnv_chain_end(h)
    only_smooth = _lib.Cfg(10, 4, [0, 10], [], stage_mask=_lib.ST_SMOOTH | _lib.ST_CENTER)
    assert L.icnv_chain_begin(ct.byref(h), only_smooth.ptr()) == _lib.OK
    assert L.icnv_chain_num_rounds(h) == 0
    L.icnv_chain_end(h)


def test_chr_layout_permutation():
    from infercnv_amd import GeneOrder, InfercnvObject
    chrs = np.array(["chr2", "chr1", "chr2", "chr3", "chr1"])
    obj = InfercnvObject(expr_data=np.zeros((5, 2)), gene_order=GeneOrder(chr=chrs))
    perm, cs = obj.chr_layout()
    assert list(chrs[perm]) == ["chr2", "chr2", "chr1", "chr1", "chr3"]   # order of first appearance
    assert list(cs) == [0, 2, 4, 5]
    obj2 = InfercnvObject(expr_data=np.zeros((4, 2)), gene_order=GeneOrder(chr=np.array(["a", "a", "b", "c"])))
    perm2, cs2 = obj2.chr_layout()
    assert perm2 is None and list(cs2) == [0, 2, 3, 4]


def test_synthetic_generator_is_deterministic_and_shardable():
    import torch
    from infercnv_amd import synth
    x, cs = synth.make_matrix_np(1000, 40)
    assert cs[0] == 0 and cs[-1] == 1000 and len(cs) == 23 and (np.diff(cs) >= 1).all()
    y, _ = synth.make_matrix_np(1000, 10, cell_offset=20, C_total=40)
    np.testing.assert_array_equal(y, x[:, 20:30])
    t, _ = synth.make_matrix_torch(1000, 40, "cpu")
    assert np.abs(t.numpy().T - x).max() < 1e-13
    assert list(synth.chr_layout(10000)[:3]) == [0, 1072, 1780]
    refs, obs = synth.groups(1000)
    assert sum(map(len, refs)) == 100 and sum(map(len, obs)) == 900


def test_shard_helpers():
    from infercnv_amd import sharded
    b = [sharded.shard_bounds(10, 3, r) for r in range(3)]
    assert b == [(0, 4), (4, 7), (7, 10)]
    loc = sharded.localize_groups([np.array([9, 0, 5, 4]), np.array([1])], 4, 7)
    assert [list(v) for v in loc] == [[1, 0], []]
    assert list(sharded.cyclic_cells(10, 3, 1)) == [1, 4, 7]
    cyc = sharded.localize_groups_cyclic([np.array([9, 0, 5, 4]), np.array([1, 7])], 1, 3)
    assert [list(v) for v in cyc] == [[1], [0, 2]]          # global 4 -> local 1; globals 1, 7 -> locals 0, 2
    cuts = sharded.align_to_groups(100, 4, [0, 30, 45, 80, 100])
    assert cuts == [(0, 30), (30, 45), (45, 80), (80, 100)]


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "oracle")); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import oracle_np as onp
from parity_util import check_denoise_flips
from infercnv_amd import sharded, synth

class OracleEngine:
    """Same four methods as device.ChainPlan, arithmetic by the NumPy oracle (test-only)."""
    def __init__(self, G, chr_codes, ref_groups_local):
        self.G, self.chr_codes, self.refs = G, chr_codes, ref_groups_local
        self.num_rounds = 3
        self.b1 = self.b2 = self.den = None
    def _upto(self, x, stage):
        v = x
        if stage >= 1: v = onp.subtract_expr(v, self.b1, True); v = onp.apply_max_threshold_bounds(v, 3.0)
        if stage >= 1: v = onp.center_columns(onp.smooth_by_chromosome(v, self.chr_codes, 101))
        if stage >= 2: v = onp.invert_log2(onp.subtract_expr(v, self.b2, True))
        return v
    def round_partial(self, r, x):
        v = self._upto(x, r)
        if r < 2:
            sums = [v[:, g].sum(axis=1) if len(g) else np.zeros(self.G) for g in self.refs]
            self.buf = torch.from_numpy(np.concatenate(sums + [np.array([float(len(g)) for g in self.refs])]))
        else:
            idx = np.concatenate(self.refs).astype(int)
            vals = v[:, idx]
            sd = vals.std(axis=0, ddof=1).sum() if idx.size else 0.0
            self.buf = torch.tensor([vals.sum(), sd, float(idx.size), float(idx.size * self.G)], dtype=torch.float64)
        return self.buf
    def round_finish(self, r):
        b = self.buf.numpy()
        if r < 2:
            n = len(self.refs)
            means = (b[:self.G * n].reshape(n, self.G) / b[self.G * n:][:, None]).T
            if r == 0: self.b1 = means
            else: self.b2 = means
        else:
            self.den = (b[0] / b[3], b[1] / b[2] * 1.5)
    def apply(self, x, out=None, want_pre_denoise=False):
        pre = self._upto(x, 2)
        return onp.clear_noise_bounds(pre, *self.den), pre

dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
G, C = 600, 41
x, cs = synth.make_matrix_np(G, C)
refs, _ = synth.groups(C, ref_frac=0.3)
refs = [refs[0][::-1].copy(), refs[1]]                      # unsorted group, spans both shards
refs[1] = np.concatenate([refs[1], [C - 1]]).astype(np.int32)
chr_codes = np.repeat(np.arange(len(cs) - 1), np.diff(cs))
c0, c1 = sharded.shard_bounds(C, 2, rank)
eng = OracleEngine(G, chr_codes, sharded.localize_groups(refs, c0, c1))
out, pre = sharded.ShardedChain(eng).run(np.ascontiguousarray(x[:, c0:c1]), want_pre_denoise=True)
want, want_pre = onp.run_chain(x, chr_codes, refs, return_pre_denoise=True)
assert np.abs(pre - want_pre[:, c0:c1]).max() < 1e-12, np.abs(pre - want_pre[:, c0:c1]).max()
mu, s = onp.clear_noise_params_via_ref_mean_sd(want_pre, np.concatenate(refs), 1.5)
check_denoise_flips(out, want[:, c0:c1], want_pre[:, c0:c1], mu, s, tol=1e-12, label=f"gloo rank {rank}, blocks")
# the same run with the cells dealt round-robin (what bench.py --gpus N does)
mine = sharded.cyclic_cells(C, 2, rank)
eng = OracleEngine(G, chr_codes, sharded.localize_groups_cyclic(refs, rank, 2))
out, pre = sharded.ShardedChain(eng).run(np.ascontiguousarray(x[:, mine]), want_pre_denoise=True)
assert np.abs(pre - want_pre[:, mine]).max() < 1e-12, np.abs(pre - want_pre[:, mine]).max()
check_denoise_flips(out, want[:, mine], want_pre[:, mine], mu, s, tol=1e-12, label=f"gloo rank {rank}, round-robin")
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_sharded_chain_two_ranks_gloo(tmp_path):
    """world_size-2 run of the N>1 orchestration (reference rounds + all-reduce) on CPU."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(port), str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
        assert f"rank {r} ok" in o


WORKER8 = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "oracle")); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import oracle_np as onp, oracle_c as oc
from parity_util import check_denoise_flips
from infercnv_amd import sharded, synth
oc.set_num_threads(1)
W = int(sys.argv[4])

class Engine:
    """device.ChainPlan's four methods: the expensive stages by the C oracle (one thread), the elementwise ones in NumPy;
    the partial statistics are plain fp64 sums over THIS rank's reference cells -- what the all-reduce then adds in rank order."""
    def __init__(self, G, cs, refs_local):
        self.G, self.cs, self.refs, self.num_rounds = G, cs, refs_local, 3
        self.b1 = self.b2 = self.den = None
    def _upto(self, x, stage):
        v = x
        if stage >= 1:
            v = onp.apply_max_threshold_bounds(onp.subtract_expr(v, self.b1, True), 3.0)
            v = oc.center_columns(oc.smooth_by_chromosome(v, self.cs, 101), "median")
        if stage >= 2: v = onp.invert_log2(onp.subtract_expr(v, self.b2, True))
        return v
    def round_partial(self, r, x):
        v = self._upto(x, r)
        if r < 2:
            sums = [v[:, g].sum(axis=1) if len(g) else np.zeros(self.G) for g in self.refs]
            self.buf = torch.from_numpy(np.concatenate(sums + [np.array([float(len(g)) for g in self.refs])]))
        else:
            idx = np.concatenate(self.refs).astype(int)
            vals = v[:, idx]
            self.buf = torch.tensor([vals.sum(), vals.std(axis=0, ddof=1).sum() if idx.size else 0.0, float(idx.size), float(idx.size * self.G)], dtype=torch.float64)
        return self.buf
    def round_finish(self, r):
        b = self.buf.numpy()
        if r < 2:
            n = len(self.refs)
            m = (b[:self.G * n].reshape(n, self.G) / b[self.G * n:][:, None]).T
            if r == 0: self.b1 = m
            else: self.b2 = m
        else:
            self.den = (b[0] / b[3], b[1] / b[2] * 1.5)
    def apply(self, x, out=None, want_pre_denoise=False):
        pre = self._upto(x, 2)
        return onp.clear_noise_bounds(pre, *self.den), pre

dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=W)
rank = dist.get_rank()
G, C = 2000, 4000
cs = synth.chr_layout(G)
mine = sharded.cyclic_cells(C, W, rank)                               # the round-robin deal of bench.py --gpus N
x_mine = np.concatenate([synth.make_matrix_np(G, 1, cell_offset=int(c), C_total=C)[0] for c in mine], axis=1)
refs, _ = synth.groups(C)
eng = Engine(G, cs, sharded.localize_groups_cyclic(refs, rank, W))
out, pre = sharded.ShardedChain(eng).run(x_mine, want_pre_denoise=True)
# the ONE-rank run of the same engine on the same cells: the reference cells' statistics summed in one go (no rank order)
ref_all = np.concatenate(refs)
x_ref = np.concatenate([synth.make_matrix_np(G, 1, cell_offset=int(c), C_total=C)[0] for c in ref_all], axis=1)
off = np.concatenate([[0], np.cumsum([len(r) for r in refs])])
one = Engine(G, cs, [np.arange(off[i], off[i + 1]) for i in range(len(refs))])
for r in range(3):
    one.round_partial(r, x_ref); one.round_finish(r)
one_out, one_pre = one.apply(x_mine)
d_pre = float(np.abs(pre - one_pre).max())
assert d_pre < 1e-12, d_pre                                             # sums in another order: rounding only
mu, s = one.den
flips = check_denoise_flips(out, one_out, one_pre, mu, s, tol=1e-12, label=f"{W} ranks vs one, rank {rank}")
means, sd, logPi, logDelta = synth.hmm_params_i6()
st, _ = oc.viterbi_cells(pre, cs, means, sd, logPi, logDelta)
st1, _ = oc.viterbi_cells(one_pre, cs, means, sd, logPi, logDelta)
mism = int((st != st1).sum())
tot = torch.tensor([float(flips), float(mism), float(out.size), d_pre], dtype=torch.float64)
mx = tot.clone()
dist.all_reduce(tot, op=dist.ReduceOp.SUM)
dist.all_reduce(mx, op=dist.ReduceOp.MAX)
if rank == 0:
    print(f"WORLD{W}: {int(tot[0])} legal step-22 flips and {int(tot[1])} differing state calls in {int(tot[2])} elements; max |pre - pre_1rank| {float(mx[3]):.2e}")
assert mism == 0
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_sharded_chain_eight_ranks_vs_one_rank_gloo(tmp_path):
    """What N ranks change against one: the reference statistics are all-reduced -- partial sums added in rank order --, so the
    bounds of steps 8 / 12 and the step-22 parameters differ from the one-rank sums in their last bits.  Eight gloo ranks
    (cells dealt round-robin like bench.py --gpus 8) against the same engine on one rank: the pre-denoise matrix within
    1e-12, every step-22 difference a legal flip on a bound (counted and printed: 0 expected at 8e6 elements, 0-3 per 1e8),
    the i6 state calls IDENTICAL."""
    script = tmp_path / "worker8.py"
    script.write_text(WORKER8)
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    W = 8
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(port), str(r), str(W)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(W)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
        assert f"rank {r} ok" in o
    line = [l for l in outs[0].splitlines() if l.startswith("WORLD8")]
    assert line, outs[0][-2000:]
    print(line[0])


def test_median_network_header_is_generated_and_verified():
    """median9x9_net.h (the min/max networks of the 9 x 9 median filter) is exactly what gen_median_net.py writes; the
    generator checks every network first: sort9 exhaustively (0-1 principle), the single-output, two-rank (padded
    border windows) and shared-window pair networks on thousands of random inputs with ties."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "infercnv_amd", "csrc", "gen_median_net.py"), "--check"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "up to date" in r.stdout


def test_median_strip_network_header_is_generated_and_verified():
    """median9_strip_net.h (round 6: the 32-bit min/max networks of the strip kernel -- sort9, 9 + 9, 18 + 18, the pruned 36 + 36 window and the
    three-rank finish) is exactly what gen_median_strip_net.py writes; the generator checks every network on thousands of random inputs
    with ties and the whole sliding schedule (rows -> pairs -> quads -> window -> ranks 39, 40, 41) against sorted windows."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "infercnv_amd", "csrc", "gen_median_strip_net.py"), "--check"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "up to date" in r.stdout


def _build_shim_driver():
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mock = os.path.join(root, "rglue", "mock")
    res = subprocess.run(["make", "-C", mock, "syntax", "all"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    return os.path.join(mock, "test_shim")


def test_r_shim_compiles_and_registers():
    """rglue/src/icnv_shim.c (the .Call shim a maintainer drops into the package; R itself is not in the image) goes
    through gcc -Wall -Wextra -Werror against a mock of the R API subset it uses, its registration table is checked
    against the routines' real arities (function types at compile time, numArgs at run time), and the error path --
    the library has no device here, returns a code, the shim raises Rf_error after the library returned -- is driven
    from C (rglue/mock/test_shim.c).  The routines' results are checked on the GPU (tests/test_gpu_entrypoints.py)."""
    import subprocess
    exe = _build_shim_driver()
    res = subprocess.run([exe, "cpu"], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1"))
    assert res.returncode == 0 and "SHIM_CPU_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


WORKER_GROUPS = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "oracle"))
import numpy as np, torch, torch.distributed as dist
import oracle_np as onp
from infercnv_amd import sharded, synth

class OracleGroupEngine:
    """The three methods of sharded.DeviceGroupEngine, arithmetic by the NumPy oracle (test-only)."""
    def __init__(self, chr_codes): self.chr_codes = chr_codes
    def moments_partial(self, x, ref, phase, mean):
        v = x[:, np.asarray(ref, dtype=np.int64)].astype(np.longdouble)
        return (float(v.sum()), float(v.size)) if phase == 0 else (float(((v - np.longdouble(mean)) ** 2).sum()), float(v.size))
    def viterbi_groups(self, x, chr_start, groups, means, sds, logPi, logDelta):
        return onp.predict_cnv_on_groups(x, self.chr_codes, groups, means, [np.full(len(means), s) for s in sds], np.exp(logPi), np.exp(logDelta))
    def median_filter(self, x, chr_start, tiles, window):
        return onp.apply_median_filtering(x, self.chr_codes, tiles, window)

dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
G, C = 700, 230
x, cs = synth.make_matrix_np(G, C)
chr_codes = np.repeat(np.arange(len(cs) - 1), np.diff(cs))
rng = np.random.default_rng(5)
perm = rng.permutation(C)
sizes = [40, 7, 55, 1, 30, 23, 50, 24]                     # subclusters of uneven size, cells scattered over the matrix
off = np.concatenate([[0], np.cumsum(sizes)])
groups = [perm[off[i]:off[i + 1]] for i in range(len(sizes))]
is_ref = [False, False, True, False, False, True, False, False]
# config 4: i3 at subcluster level -- whole groups per rank, (mu, sigma) by two all-reduces of two doubles
owned = sharded.assign_groups(sizes, 2)[rank]
assert sorted(sharded.assign_groups(sizes, 2)[0] + sharded.assign_groups(sizes, 2)[1]) == list(range(len(sizes)))
cells, local = sharded.gather_groups(groups, owned)
ref_local = np.concatenate([local[j] for j, gid in enumerate(owned) if is_ref[gid]] or [np.zeros(0, dtype=np.int32)])
xl = np.ascontiguousarray(x[:, cells])
hmm = sharded.ShardedGroupHMM(OracleGroupEngine(chr_codes))
mu, sigma, delta = hmm.i3_params(xl, ref_local, 0.05)
ref_all = np.concatenate([g for g, r in zip(groups, is_ref) if r])
wmu, wsigma, wdelta = onp.i3_params(x, ref_all, 0.05)
assert abs(mu - wmu) < 1e-14 and abs(sigma - wsigma) < 1e-14 and abs(delta - wdelta) < 1e-13, (mu, wmu, sigma, wsigma)
st = hmm.run_i3(xl, cs, local, ref_local)
Pi3, d3 = onp.get_HMM_i3(1e-6)
want = onp.predict_cnv_on_groups(x, chr_codes, groups, np.array([wmu - wdelta, wmu, wmu + wdelta]), [np.full(3, wsigma)] * len(groups), Pi3, d3)
assert np.array_equal(st, want[:, cells]), (st != want[:, cells]).sum()
# config 5: the median filter on whole tiles per rank -- no collective
mf = sharded.ShardedMedianFilter(OracleGroupEngine(chr_codes)).run(xl, cs, local, 7)
wmf = onp.apply_median_filtering(x, chr_codes, groups, 7)
assert np.array_equal(mf, wmf[:, cells])
# contiguous-block groups: cuts moved to group boundaries
bounds = sharded.align_to_groups(230, 2, off)
assert bounds[0][1] in off and bounds[0][1] == bounds[1][0]
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_sharded_group_hmm_and_median_filter_two_ranks_gloo(tmp_path):
    """BASELINE configs 4 and 5 on two ranks (gloo, CPU, oracle-backed engine): whole groups / tiles per rank
    (assign_groups + gather_groups), the i3 (mu, sigma) from the split-phase moments by two all-reduces of two doubles
    (SURVEY.md 8e), group HMM states and median-filter output of every rank's cells equal to the one-rank result."""
    script = tmp_path / "worker_groups.py"
    script.write_text(WORKER_GROUPS)
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(port), str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
        assert f"rank {r} ok" in o


WORKER_INGEST = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "oracle"))
import numpy as np, torch, torch.distributed as dist
import oracle_np as onp
from infercnv_amd import sharded

class NumpyIngestEngine:
    """The four methods ShardedIngest needs, in NumPy on integer counts (test-only stand-in for device.ingest_*)."""
    @staticmethod
    def gene_stats(c): return torch.from_numpy(np.concatenate([c.sum(axis=1), (c > 0).sum(axis=1)]).astype(np.float64))
    @staticmethod
    def select(st, G, C_total, cutoff, min_cells):
        keep = np.ones(G, dtype=bool)
        if cutoff is not None: keep &= ~(st[:G] / C_total < cutoff)
        if min_cells > 0: keep &= st[G:] >= min_cells
        return np.nonzero(keep)[0].astype(np.int32)
    @staticmethod
    def col_sums(c, keep): return torch.from_numpy(c[keep].sum(axis=0).astype(np.float64))
    @staticmethod
    def apply(c, keep, cs, factor): return np.log2(c[keep].astype(np.float64) / cs.numpy() * factor + 1.0)

dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
rng = np.random.default_rng(9)
G, C = 500, 91                                              # odd number of cells: the ranks' blocks differ in size
counts = rng.poisson(rng.gamma(0.3, 4.0, size=(G, 1)), size=(G, C)).astype(np.int64)
c0, c1 = sharded.shard_bounds(C, 2, rank)
out, keep, factor = sharded.ShardedIngest(NumpyIngestEngine).run(counts[:, c0:c1], C, 0.5, 3)
f = counts.astype(np.float64)
drop = onp.below_min_mean_expr_cutoff(f, 0.5)
k = np.setdiff1d(np.arange(G), drop)
k = k[onp.genes_passing_min_cells(f[k], 3)]
assert np.array_equal(keep, k) and 50 < k.size < G
want = onp.log2xplus1(onp.normalize_counts_by_seq_depth(f[k]))
assert factor == float(np.median(f[k].sum(axis=0)))
assert np.abs(out - want[:, c0:c1]).max() < 1e-12
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_sharded_ingest_two_ranks_gloo(tmp_path):
    """Steps 2-4 from integer counts on two ranks (gloo, CPU, NumPy engine): gene statistics all-reduced, the filter
    decision identical on both ranks, column sums all-gathered for the global median (ranks of different size) -- the
    rank's block equals the one-rank result of the reference's four step functions (oracle)."""
    script = tmp_path / "worker_ingest.py"
    script.write_text(WORKER_INGEST)
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(port), str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
        assert f"rank {r} ok" in o


# ------------------------------------------------------------------ R glue: .Call arity against the shim's registration table
def _r_call_sites(text):
    """Every `.Call("name", arg, ...)` of an R source: (name, number of arguments after the name, line).  Arguments are
    split at top-level commas (parentheses / brackets / braces and string literals respected, `#` comments dropped)."""
    clean = []
    for line in text.split("\n"):
        out, q = [], None
        for ch in line:
            if q:
                out.append(ch)
                if ch == q:
                    q = None
            elif ch in "\"'":
                q = ch
                out.append(ch)
            elif ch == "#":
                break
            else:
                out.append(ch)
        clean.append("".join(out))
    src = "\n".join(clean)
    sites = []
    for m in re.finditer(r"\.Call\(", src):
        i, depth, q, parts, cur = m.end(), 1, None, [], []
        while i < len(src) and depth:
            ch = src[i]
            if q:
                cur.append(ch)
                if ch == q:
                    q = None
            elif ch in "\"'":
                q = ch
                cur.append(ch)
            elif ch in "([{":
                depth += 1
                cur.append(ch)
            elif ch in ")]}":
                depth -= 1
                if depth:
                    cur.append(ch)
            elif ch == "," and depth == 1:
                parts.append("".join(cur).strip())
                cur = []
            else:
                cur.append(ch)
            i += 1
        parts.append("".join(cur).strip())
        assert depth == 0, "unbalanced .Call"
        name = parts[0].strip("\"'")
        sites.append((name, len(parts) - 1, src.count("\n", 0, m.start()) + 1))
    return sites


def test_r_wrappers_call_the_shim_with_the_registered_arity():
    """rglue/R/zzz_hip_backend.R cannot be parsed by R in this image (no R): what CAN be checked statically is that every
    `.Call("icnv_R_x", ...)` names a routine the shim registers (rglue/src/icnv_shim.c, `call_methods[]`) and passes exactly
    as many arguments as the registration declares -- R would stop with "Incorrect number of arguments" at the first call
    otherwise -- and that every registered routine is reached from the wrappers."""
    rsrc = open(os.path.join(ROOT, "rglue", "R", "zzz_hip_backend.R")).read()
    csrc = open(os.path.join(ROOT, "rglue", "src", "icnv_shim.c")).read()
    table = csrc[csrc.index("call_methods[]"):]
    table = table[:table.index("};")]
    registered = {m.group(1): int(m.group(2)) for m in re.finditer(r'\{"(icnv_R_\w+)",\s*\(DL_FUNC\)&\w+,\s*(\d+)\}', table)}
    assert len(registered) >= 10
    # the C definitions themselves: SEXP icnv_R_x(SEXP a, SEXP b, ...) has as many parameters as it registers
    for name, n in registered.items():
        m = re.search(r"SEXP\s+" + name + r"\s*\(([^)]*)\)", csrc)
        assert m, name
        params = [p for p in m.group(1).split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == n, (name, len(params), n)
    sites = _r_call_sites(rsrc)
    assert sites, "no .Call in the R wrappers?"
    for name, nargs, line in sites:
        assert name in registered, f"zzz_hip_backend.R:{line}: .Call to unregistered routine {name}"
        assert nargs == registered[name], f"zzz_hip_backend.R:{line}: {name} called with {nargs} arguments, registered with {registered[name]}"
    assert {s[0] for s in sites} == set(registered), sorted(set(registered) - {s[0] for s in sites})


def test_r_wrappers_are_lexically_well_formed():
    """The least a file R has never parsed must satisfy: with string literals and comments removed, every ( [ { closes in
    the right order, and every step function of SURVEY 8b.1 is (re)bound to a `function(...)`."""
    rsrc = open(os.path.join(ROOT, "rglue", "R", "zzz_hip_backend.R")).read()
    stack, q, line = [], None, 1
    pairs = {")": "(", "]": "[", "}": "{"}
    i = 0
    while i < len(rsrc):
        ch = rsrc[i]
        if ch == "\n":
            line += 1
        if q:
            if ch == "\\":
                i += 1
            elif ch == q:
                q = None
        elif ch in "\"'":
            q = ch
        elif ch == "#":
            while i < len(rsrc) and rsrc[i] != "\n":
                i += 1
            continue
        elif ch in "([{":
            stack.append((ch, line))
        elif ch in ")]}":
            assert stack and stack[-1][0] == pairs[ch], f"zzz_hip_backend.R:{line}: unmatched {ch}"
            stack.pop()
        i += 1
    assert not stack and q is None, stack[-3:]
    for fn in ("subtract_ref_expr_from_obs", "apply_max_threshold_bounds", "smooth_by_chromosome", "center_cell_expr_across_chromosome",
               "invert_log2", "clear_noise_via_ref_mean_sd", "clear_noise", "predict_CNV_via_HMM_on_indiv_cells",
               "predict_CNV_via_HMM_on_tumor_subclusters", "predict_CNV_via_HMM_on_tumor_subclusters_per_chr",
               "predict_CNV_via_HMM_on_whole_tumor_samples", "i3HMM_predict_CNV_via_HMM_on_indiv_cells",
               "i3HMM_predict_CNV_via_HMM_on_tumor_subclusters", "i3HMM_predict_CNV_via_HMM_on_whole_tumor_samples",
               "assign_HMM_states_to_proxy_expr_vals", "i3HMM_assign_HMM_states_to_proxy_expr_vals", "apply_median_filtering"):
        assert re.search(r"\b" + re.escape(fn) + r"\b", rsrc), fn


def test_bench_self_launch_starts_n_ranks_with_every_flag_passed_through(monkeypatch):
    """`python bench.py --gpus N` without a launcher must not die at argv (the driver's scaling run may be called exactly
    like its single-GPU one): bench.self_launch re-executes the same command line under torch.distributed.run with N
    ranks, a free port on 127.0.0.1, every flag passed through, and returns the launcher's exit code.  With WORLD_SIZE set
    (a real launcher) main() must NOT launch again.  No GPU involved: subprocess.run is intercepted."""
    import importlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1", "--config", "3"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                     # the launcher's exit code is bench.py's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    k = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "8", "--steps", "3", "--warmup", "1", "--config", "3"]
    assert seen["env"]["ICNV_BENCH_LAUNCHER"] == "self" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # a mismatch between --gpus and a launcher's WORLD_SIZE is an error message, not a silent one-rank run
    monkeypatch.setenv("WORLD_SIZE", "2")
    seen.clear()
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert not seen and "WORLD_SIZE is 2" in str(e.value.code)


def test_r_wrappers_take_the_formals_of_the_step_functions_they_replace():
    """Every `hip_<step function>` of rglue/R/zzz_hip_backend.R is bound over the reference's function of that name
    (.icnv_enable_hip_backend), so it must accept the same formals in the same order, with a default wherever the reference
    has one -- infercnv::run() calls them positionally and by name (R/inferCNV_ops.R:771-1589).  The reference's signatures are
    a committed fixture (tests/golden/r_step_function_signatures.json, extracted by tests/golden/make_golden.py from the
    reference's sources: names, order, default flags); the glue is read with the same parser (no R in this image).  Also: the
    backend switch re-binds exactly these functions, and each `hip_*` body closes (its formals parse at all)."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import r_signatures
    rsrc = open(os.path.join(ROOT, "rglue", "R", "zzz_hip_backend.R")).read()
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "r_step_function_signatures.json")))
    assert len(want) >= 20
    swap = dict(re.findall(r'(\w+)\s*=\s*"(hip_\w+)"', rsrc[rsrc.index("swap <- c("):]))
    for name, sig in want.items():
        got = r_signatures.formals(rsrc, "hip_" + name)
        assert got is not None, f"no hip_{name} in the R glue"
        assert [a for a, _ in got] == sig["formals"], (name, [a for a, _ in got], sig["formals"])
        for (a, has), ref_has in zip(got, sig["has_default"]):
            assert has or not ref_has, f"hip_{name}: `{a}` has a default in the reference ({sig['file']}) but not in the glue"
        assert swap.get(name) == "hip_" + name, f"{name} is not re-bound by .icnv_enable_hip_backend"
    assert set(swap) == set(want), sorted(set(swap) ^ set(want))
