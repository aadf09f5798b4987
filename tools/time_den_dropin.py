"""Drop-in cost of the k2-loss pattern (rnnp.py:91-104): numerator - denominator under autograd, cfg4 mmictc sizes;
the two-call form (eager / split denominators) and graphloss_num_den (one node, one gradient box)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import waterfall_amd as wf
from waterfall_amd import graph as G
dev = torch.device("cuda:0")
P, B, T, L = 5000, 32, 500, 64
phones = ["<eps>", "<SIL>"] + ["p%d" % i for i in range(P)]
topo = wf.Topology("mmictc", phones)
V = topo.num_nn_output
rng = np.random.default_rng(0)
seqs = [[int(x) + 2 for x in rng.integers(0, P, size=L)] for _ in range(B)]
num = wf.compile_ctc_k2_graphs(topo, seqs)
den = wf.create_fsa_vec([topo.fsa] * B)
lp = torch.log_softmax(torch.randn(B, T, V, device=dev) * 3, -1).requires_grad_(True)
seg = torch.tensor([[i, 0, T] for i in range(B)], dtype=torch.int32)
for mode in (True, False, "num_den"):
    G._EAGER_DIRECT = mode is not False
    def step():
        lp.grad = None
        d = wf.DenseFsaVec(lp, seg)
        if mode == "num_den":
            loss = wf.graphloss_num_den(num, den, d, reduction="sum")
        else:
            loss = wf.graphloss(num, d, reduction="sum") - wf.graphloss(den, d, reduction="sum")
        loss.backward()
        return loss
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): l = step()
    torch.cuda.synchronize()
    print({True: "eager denominators", False: "split denominators", "num_den": "graphloss_num_den"}[mode], "%.3f ms/step" % ((time.perf_counter() - t0) / 10 * 1e3), float(l.detach()), float(lp.grad.abs().sum()))
