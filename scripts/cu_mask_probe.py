"""Would the step gain from running the fused smooth pass and the Viterbi CONCURRENTLY on disjoint halves of the CUs?
Times chain_apply and the Viterbi (i) alone on all CUs, (ii) alone on a CU-masked stream (half the CUs), (iii) together on
complementary masks.  hipExtStreamCreateWithCUMask through ctypes; the library's *_dev entry points take the stream.
A measurement script: outputs are not checked (two streams share the library's workspace pool).
    python scripts/cu_mask_probe.py [pattern]     pattern: halves (default) | interleaved | xcd
"""
import ctypes as ct
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    pattern = sys.argv[1] if len(sys.argv) > 1 else "halves"
    import torch
    from infercnv_amd import device, sharded, synth
    torch.cuda.set_device(0)
    device.init(0)
    hip = ct.CDLL("libamdhip64.so")
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (n_cu + 31) // 32

    def mask(bits):
        arr = (ct.c_uint32 * words)()
        for b in bits:
            arr[b // 32] |= 1 << (b % 32)
        s = ct.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ct.byref(s), words, arr)
        assert rc == 0, rc
        return torch.cuda.ExternalStream(s.value)

    if pattern == "halves":
        a_bits, b_bits = range(0, n_cu // 2), range(n_cu // 2, n_cu)
    elif pattern == "interleaved":
        a_bits, b_bits = range(0, n_cu, 2), range(1, n_cu, 2)
    else:   # whole XCDs alternately (bit i -> XCD i % 8 in the runtime's enumeration)
        a_bits = [i for i in range(n_cu) if (i % 8) < 4]
        b_bits = [i for i in range(n_cu) if (i % 8) >= 4]
    sa, sb = mask(a_bits), mask(b_bits)
    G, C = 10000, 25000
    means, sd, logPi, logDelta = synth.hmm_params_i6()
    halves = []
    for h in range(2):
        x, cs = synth.make_matrix_torch(G, C, "cuda", cell_offset=h * C, C_total=2 * C)
        refs = [torch.arange(0, 1250).numpy().astype("int32"), torch.arange(1250, 2500).numpy().astype("int32")]
        out = torch.empty_like(x)
        st = torch.empty((C, G), dtype=torch.uint8, device="cuda")
        plan = device.ChainPlan(G, C, cs, refs)
        chain = sharded.ShardedChain(plan)
        _, pre = chain.run(x, out=out, want_pre_denoise=True)
        pre = pre.clone()
        halves.append(dict(x=x, cs=cs, out=out, st=st, plan=plan, pre=pre))
    torch.cuda.synchronize()

    def chain_apply(h, stream):
        with torch.cuda.stream(stream):
            h["plan"].apply(h["x"], out=h["out"], want_pre_denoise=True)

    def vit(h, stream):
        with torch.cuda.stream(stream):
            device.viterbi_cells(h["pre"], h["cs"], means, sd, logPi, logDelta, states=h["st"])

    def timed(label, fn, reps=10):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        print("%-70s %.3f ms" % (label, (time.perf_counter() - t0) / reps * 1e3))

    d = torch.cuda.default_stream()
    A, B = halves
    # the plan's reference statistics are in place (chain.run above); apply() recomputes the reference cells when the cache is gone
    timed("chain_apply, 25 000 cells, all CUs", lambda: chain_apply(A, d))
    timed("viterbi, 25 000 cells, all CUs", lambda: vit(A, d))
    timed("chain_apply(A) then viterbi(B), all CUs, one stream", lambda: (chain_apply(A, d), vit(B, d)))
    timed("chain_apply, 25 000 cells, mask A (half the CUs)", lambda: chain_apply(A, sa))
    timed("viterbi, 25 000 cells, mask B (half the CUs)", lambda: vit(B, sb))
    timed("chain_apply(A) on mask A || viterbi(B) on mask B", lambda: (chain_apply(A, sa), vit(B, sb)))
    timed("chain_apply(A) || viterbi(B), two unmasked streams", lambda: (chain_apply(A, torch.cuda.Stream()), vit(B, d)))


if __name__ == "__main__":
    main()
