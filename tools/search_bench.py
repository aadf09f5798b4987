"""Timing of wf_search (beam-pruned) against the exact sweep on the cfg5 workloads (bench.py::viterbi_bench), and the
memory check on a 10^7-arc synthetic graph.  Usage: python tools/search_bench.py [S ...]"""
import sys
import time

import torch

import waterfall_amd as wf
from waterfall_amd.synth import cfg5_workload

dev = torch.device("cuda:0")


def timed(f, reps=3):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


import itertools
for S, peaked in itertools.product([int(x) for x in sys.argv[1:]] or [100_000, 1_000_000], (True, False)):
    B, T, V = 8, 500, 5001
    if S > 1_000_000 and not peaked:
        continue
    g, fsa, lp, dense = cfg5_workload(dev, S, B, T, V, peaked=peaked and S <= 1_000_000)
    if S > 1_000_000:   # (no exact sweep to plant a path with: plant the labels of the state chain 0, 1, 2, ...)
        z = torch.randn(B, T, V, device=dev)
        nxt = [int(a) for a in range(g["src"].size) if False]
        lp = torch.log_softmax(z * 3.0, -1)
    print(f"S={S} A={fsa.arcs.shape[0]} peaked={peaked}", flush=True)
    if S <= 1_000_000:
        ms, exact = timed(lambda: wf.best_path(fsa, dense))
        print(f"  exact sweep           {ms:8.2f} ms", flush=True)
    else:
        exact = None
    for beam, mx in [(float("inf"), 0), (20.0, 10000), (12.0, 10000), (8.0, 10000), (20.0, 2000)]:
        if beam == float("inf") and (S > 100_000 or peaked):
            continue
        ms, out = timed(lambda: wf.pruned_search(fsa, dense, beam, 30, mx, log_tot=False, return_active=True))
        ms2, _ = timed(lambda: wf.pruned_search(fsa, dense, beam, 30, mx, log_tot=True))
        na = out["n_active"].float().mean().item()
        same = "-" if exact is None else str(bool(torch.equal(out["arcs"], exact["arcs"])))
        gap = "-" if exact is None else "%.4f" % (exact["scores"] - out["scores"]).max().item()
        print(f"  beam {beam:5} max_active {mx:6d}: {ms:8.2f} ms (with log total {ms2:8.2f}), mean active {na:9.1f}, "
              f"same path {same}, score gap {gap}", flush=True)
    print(f"  peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)
