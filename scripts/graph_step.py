"""The bench step (3 reference rounds + fused apply + i6 Viterbi, one GPU) launched call by call against the same step
replayed from a captured hipGraph (torch.cuda.CUDAGraph on the stream the library is handed).  Says what the host-side
launch path costs: the step is ~10 library calls and ~14 kernels of 5-2 400 us.

    python scripts/graph_step.py [--cells 50000] [--steps 20]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genes", type=int, default=10000)
    ap.add_argument("--cells", type=int, default=50000)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import torch
    from infercnv_amd import device, sharded, synth

    torch.cuda.set_device(0)
    device.init(0)
    G, C = args.genes, args.cells
    x, chr_start = synth.make_matrix_torch(G, C, "cuda")
    refs, _ = synth.groups(C)
    means, sd, logPi, logDelta = synth.hmm_params_i6()
    out = torch.empty_like(x)
    states = torch.empty((C, G), dtype=torch.uint8, device="cuda")
    chain = sharded.ShardedChain(device.ChainPlan(G, C, chr_start, refs))
    keep = {}

    def step():
        _, pre = chain.run(x, out=out, want_pre_denoise=True)
        device.viterbi_cells(pre, chr_start, means, sd, logPi, logDelta, states=states)
        keep["pre"] = pre

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    for _ in range(3):
        step()
    eager = timed(step, args.steps)
    ref = (float(out.sum()), float(keep["pre"].sum()), int(states.sum(dtype=torch.int64)))

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            step()
    except Exception as e:                                   # noqa: BLE001
        print("capture failed:", repr(e)[:300])
        print("eager %.3f ms per step" % eager)
        return
    out.zero_(); states.zero_()
    graph = timed(g.replay, args.steps)
    got = (float(out.sum()), float(keep["pre"].sum()), int(states.sum(dtype=torch.int64)))
    eager2 = timed(step, args.steps)
    print("eager %.3f ms, graph replay %.3f ms, eager again %.3f ms per step; outputs equal: %s" % (eager, graph, eager2, got == ref))


if __name__ == "__main__":
    main()
