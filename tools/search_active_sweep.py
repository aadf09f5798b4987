"""wf_search at realistic active-state counts: cfg5's S = 10^6 graph, B = 8, T = 500, search_beam 20,
the planted path's logit boost chosen so that the mean number of active states is about 500 / 2 000 / 5 000.
Usage: [SWEEP_B=1] python tools/search_active_sweep.py [peak ...]"""
import sys
import time
sys.path.insert(0, ".")
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


import os
S, B, T, V = 1_000_000, int(os.environ.get("SWEEP_B", "8")), 500, 5001   # (SWEEP_B=1: the reference's lattice_analysis.py decodes one utterance at a time)
for peak in [float(x) for x in sys.argv[1:]] or [10.0, 6.0, 5.0, 4.0, 3.0]:
    g, fsa, lp, dense = cfg5_workload(dev, S, B, T, V, peaked=True, peak=peak)
    ms_exact, exact = timed(lambda: wf.best_path(fsa, dense))
    for beam, mx in [(20.0, 10000), (12.0, 10000)]:
        ms, out = timed(lambda: wf.pruned_search(fsa, dense, beam, 30, mx, log_tot=False, return_active=True))
        ms2, _ = timed(lambda: wf.pruned_search(fsa, dense, beam, 30, mx, log_tot=True))
        ms3, lat = timed(lambda: wf.pruned_search(fsa, dense, beam, 30, mx, log_tot=True, output_beam=8.0))
        na = out["n_active"].float().mean().item()
        print(f"peak {peak:4.1f} beam {beam:4.0f}: {ms:7.2f} ms (log total {ms2:7.2f}, + output_beam 8 {ms3:7.2f}, {lat['lattice_arcs'].double().mean().item() / T:7.0f} arcs/frame) exact {ms_exact:6.2f} ms  x{ms_exact / ms:5.2f}  "
              f"mean active {na:8.1f} max {out['n_active'].max().item():6d}  same path {bool(torch.equal(out['arcs'], exact['arcs']))}", flush=True)
