"""Do the step's two hot kernels share a clock / power budget?  Times the fused smooth pass and the Viterbi (i) each on its
own, back to back with itself, (ii) alternating as in the bench step, (iii) alternating with a pause in front of each
launch.  Per-kernel HIP-event durations from the library's timers.

    python scripts/power_coupling.py [reps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    import torch
    from infercnv_amd import device, sharded, synth
    torch.cuda.set_device(0)
    device.init(0)
    G, C = 10000, 50000
    x, cs = synth.make_matrix_torch(G, C, "cuda")
    refs, _ = synth.groups(C)
    means, sd, logPi, logDelta = synth.hmm_params_i6()
    out = torch.empty_like(x)
    states = torch.empty((C, G), dtype=torch.uint8, device="cuda")
    plan = device.ChainPlan(G, C, cs, refs)
    chain = sharded.ShardedChain(plan)
    _, pre = chain.run(x, out=out, want_pre_denoise=True)

    def chain_only():
        for r in range(plan.num_rounds):
            plan.round_partial(r, x)
            plan.round_finish(r)
        plan.apply(x, out=out, want_pre_denoise=True)

    def vit_only():
        device.viterbi_cells(pre, cs, means, sd, logPi, logDelta, states=states)

    def measure(label, fns, pause=0.0):
        for f in fns:
            f()
        torch.cuda.synchronize()
        device.timing_reset()
        device.timing_enable(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            for f in fns:
                if pause:
                    torch.cuda.synchronize()
                    time.sleep(pause)
                f()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        device.timing_enable(False)
        ca, na = device.timing_get("chain_apply")
        vi, nv = device.timing_get("viterbi")
        print("%-52s chain_apply %s  viterbi %s  wall per round %.3f ms" % (
            label, ("%.3f ms" % (ca / na)) if na else "   -    ", ("%.3f ms" % (vi / nv)) if nv else "   -    ", wall))

    measure("smooth pass alone, back to back", [chain_only])
    measure("Viterbi alone, back to back", [vit_only])
    measure("alternating (the bench step)", [chain_only, vit_only])
    measure("alternating, 20 ms idle in front of every launch", [chain_only, vit_only], pause=0.02)
    measure("Viterbi alone again", [vit_only])
    measure("smooth pass alone again", [chain_only])


if __name__ == "__main__":
    main()
