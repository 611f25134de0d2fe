"""What the three all-reduces of a step cost on the software side: the bench step on one GPU with the reference statistics
pushed through torch.distributed all_reduce (backend nccl = RCCL, world size 1: the collective is a local copy, what
remains is the enqueue path -- stream hand-over to RCCL's stream and back, three times per step) against the same step
without them.  The link time of a 160 KB all-reduce over xGMI comes on top on a real multi-GPU node (the driver's run).

    python scripts/allreduce_overhead.py [steps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    import torch
    import torch.distributed as dist
    from infercnv_amd import device, synth
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    torch.cuda.set_device(0)
    device.init(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    G, C = 10000, 50000
    x, cs = synth.make_matrix_torch(G, C, "cuda")
    refs, _ = synth.groups(C)
    means, sd, logPi, logDelta = synth.hmm_params_i6()
    out = torch.empty_like(x)
    states = torch.empty((C, G), dtype=torch.uint8, device="cuda")
    plan = device.ChainPlan(G, C, cs, refs)

    def step(reduce):
        for r in range(plan.num_rounds):
            buf = plan.round_partial(r, x)
            if reduce:
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            plan.round_finish(r)
        _, pre = plan.apply(x, out=out, want_pre_denoise=True)
        device.viterbi_cells(pre, cs, means, sd, logPi, logDelta, states=states)

    def timed(reduce):
        for _ in range(5):
            step(reduce)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(reduce)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    a = timed(False); b = timed(True); a2 = timed(False); b2 = timed(True)
    print("step without all-reduce %.3f / %.3f ms, with three RCCL all-reduce calls (world size 1) %.3f / %.3f ms: %.0f us per step, %.0f us per call"
          % (a, a2, b, b2, ((b + b2) - (a + a2)) / 2 * 1e3, ((b + b2) - (a + a2)) / 6 * 1e3))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
