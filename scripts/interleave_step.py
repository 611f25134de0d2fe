"""The bench step with the cells in K chunks: reference rounds over all chunks (partials added), then per chunk the fused
smooth pass followed by its Viterbi -- a memory-bound launch in front of every compute-bound one, K times per step.
Tests whether the clock headroom the Viterbi gains behind a memory-bound kernel (scripts/power_coupling.py) is worth more
than the extra launches and tails.

    python scripts/interleave_step.py [steps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    import torch
    from infercnv_amd import device, sharded, synth
    torch.cuda.set_device(0)
    device.init(0)
    G, C = 10000, 50000
    refs_global, _ = synth.groups(C)
    means, sd, logPi, logDelta = synth.hmm_params_i6()

    def build(K):
        parts = []
        for r in range(K):
            x, cs = synth.make_matrix_torch(G, C // K, "cuda", cell_offset=r, cell_stride=K, C_total=C)
            refs = sharded.localize_groups_cyclic(refs_global, r, K)
            plan = device.ChainPlan(G, C // K, cs, refs)
            parts.append((x, cs, plan, torch.empty_like(x), torch.empty((C // K, G), dtype=torch.uint8, device="cuda")))
        return parts

    def step(parts):
        for r in range(parts[0][2].num_rounds):
            bufs = [p[2].round_partial(r, p[0]) for p in parts]
            if len(bufs) > 1:
                tot = bufs[0].clone()
                for b in bufs[1:]:
                    tot += b
                for b in bufs:
                    b.copy_(tot)
            for p in parts:
                p[2].round_finish(r)
        for x, cs, plan, out, states in parts:
            _, pre = plan.apply(x, out=out, want_pre_denoise=True)
            device.viterbi_cells(pre, cs, means, sd, logPi, logDelta, states=states)

    for K in (1, 2, 4, 1, 2, 4):
        parts = build(K)
        for _ in range(4):
            step(parts)
        torch.cuda.synchronize()
        device.timing_reset(); device.timing_enable(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            step(parts)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        device.timing_enable(False)
        ca, na = device.timing_get("chain_apply"); vi, nv = device.timing_get("viterbi")
        print("K = %d chunks: %.3f ms per step; chain_apply %.3f ms, viterbi %.3f ms per step" % (K, ms, ca / steps, vi / steps))
        del parts
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
