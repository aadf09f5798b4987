"""wf_search_lattice_words at cfg5's sizes (S = 10^6, B = 8, T = 500, search_beam 20, output_beam 8): the lattice restricted
to the best path's word sequence (utils/lattice_analysis.py:137-139,161-162), with word labels on ~4 % of the arcs
(~20 words per utterance).  Usage: python tools/time_lattice_words.py [peak ...]"""
import sys
import time
sys.path.insert(0, ".")
import numpy as np
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


S, B, T, V = 1_000_000, 8, 500, 5001
for peak in [float(x) for x in sys.argv[1:]] or [10.0, 2.6, 2.3]:
    g, fsa, lp, dense = cfg5_workload(dev, S, B, T, V, peaked=True, peak=peak)
    rng = np.random.default_rng(1)
    lab = fsa.arcs[:, 2]
    fsa.aux_labels = np.where(lab >= 0, np.where(rng.random(lab.size) < 0.04, rng.integers(1, 1000, lab.size), 0), -1).astype(np.int32)
    ms_lat, lat = timed(lambda: wf.pruned_search(fsa, dense, 20.0, 30, 10000, log_tot=True, output_beam=8.0, return_active=True))
    words = lat["aux_labels"]
    ms_w, out = timed(lambda: wf.lattice_words(fsa, dense, words, 20.0, 8.0, 30, 10000, use_double_scores=True))
    ratio = torch.exp(out["tot_words"] - out["tot_log"]).mean().item()
    print(f"peak {peak:4.1f}: search + lattice {ms_lat:7.2f} ms, + restricted to the words {ms_w:7.2f} ms  "
          f"(mean active {lat['n_active'].float().mean().item():7.1f}, {np.mean([len(w) for w in words]):.1f} words per utterance, "
          f"mean P(W in lattice) / P(lattice) = {ratio:.3f})", flush=True)
