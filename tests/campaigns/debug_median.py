import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root, os.path.join(root, "tests"), os.path.join(root, "oracle")]
import numpy as np, torch
import oracle_c as oc
from infercnv_amd import device, synth
torch.cuda.set_device(0); device.init(0)
G, C = 10000, 5000
x, cs = synth.make_matrix_torch(G, C, "cuda")
refs, _ = synth.groups(C)
out, pre = device.smooth_chain(x, cs, refs, want_pre_denoise=True)
subs, _, _ = synth.subclusters(C)
tiles = [g.astype(np.int32) for g in subs]
y = device.median_filter(out, cs, tiles, 7)
torch.cuda.synchronize()
want = oc.median_filter(out.cpu().numpy().T, cs, tiles, 7)
got = y.cpu().numpy().T
bad = np.argwhere(got != want)
print("mode", os.environ.get("ICNV_MF9_MODE"), "mismatches", len(bad))
mu = np.median(out.cpu().numpy())
cell_tile = np.zeros(C, dtype=np.int64); cell_pos = np.zeros(C, dtype=np.int64)
for t, g in enumerate(tiles):
    cell_tile[g] = t; cell_pos[g] = np.arange(len(g))
chr_of = np.repeat(np.arange(22), np.diff(cs))
for g, c in bad[:40]:
    k = chr_of[g]
    print(f"gene {g} (chr {k}, rel {g - cs[k]} of {cs[k+1]-cs[k]}) cell {c} (tile {cell_tile[c]} pos {cell_pos[c]} of {len(tiles[cell_tile[c]])}) got {got[g,c]:.6f} want {want[g,c]:.6f} got==mu {got[g,c]==mu} want==mu {want[g,c]==mu}")
