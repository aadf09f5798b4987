"""Times wf.best_path / tropical tot_scores on the synthetic decoding graph of BASELINE.md cfg5.
usage: python tools/bench_viterbi.py [S] [B] [T] [V]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import util  # noqa: E402
import waterfall_amd as wf  # noqa: E402

S = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T = int(sys.argv[3]) if len(sys.argv) > 3 else 500
V = int(sys.argv[4]) if len(sys.argv) > 4 else 5001
t0 = time.time()
g = util.synth_decoding_graph(777, S, V)
arcs = np.stack([g["src"], g["dst"], g["label"], g["score"].view(np.int32)], 1).astype(np.int32)
one = wf.Fsa(arcs, g["aux"], [0, arcs.shape[0]], [g["num_states"]])
fsa = wf.create_fsa_vec([one] * B)
gen = torch.Generator().manual_seed(777)
lp = torch.log_softmax(torch.randn(B, T, V, generator=gen) * 3.0, -1).cuda()
segs = torch.tensor([[i, 0, T] for i in range(B)], dtype=torch.int32)
dense = wf.DenseFsaVec(lp, segs)
print("setup %.1f s" % (time.time() - t0), flush=True)
for name, fn in (("best_path", lambda: wf.best_path(fsa, dense)),
                 ("tot_scores(tropical)", lambda: wf.tot_scores(fsa, dense, log_semiring=False)),
                 ("tot_scores(log)", lambda: wf.tot_scores(fsa, dense, log_semiring=True))):
    t0 = time.time()
    fn()
    torch.cuda.synchronize()
    first = time.time() - t0
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    n = 3
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / n
    A = arcs.shape[0]
    byt = B * T * (4 * V + 16 * A + 4 * S)
    print("%-22s S=%d A=%d B=%d T=%d: first %.2f s, %.2f ms/call, %.1f us/frame, %.3g utt-frames/s, alg %.0f GB/s"
          % (name, S, A, B, T, first, ms, ms * 1e3 / T, B * T / ms * 1e3, byt / ms / 1e6), flush=True)
