"""Cost of the reference-literal output_beam (set_output_beam_mode("pruned")) against the exact default, through the
drop-in API: cfg3 numerators (CTC, V=5001, L=64) and cfg2 (2state, V=59, L=80) numerators + denominators, B=32, T=500."""
import sys
import time
sys.path.insert(0, ".")
import numpy as np
import torch
import waterfall_amd as wf

dev = torch.device("cuda:0")
B, T = 32, 500
for name, topo_name, P, L, den in (("cfg3 ctc numerator", "ctc", 5000, 64, False), ("cfg2 2state numerator", "2state", 29, 80, False),
                                   ("cfg2 2state denominator", "2state", 29, 80, True)):
    phones = ["<eps>"] + ["p%d" % i for i in range(P)]
    topo = wf.Topology(topo_name, phones)
    V = topo.num_nn_output
    rng = np.random.default_rng(0)
    seqs = [[int(x) + 1 for x in rng.integers(0, P, size=L)] for _ in range(B)]
    g = wf.create_fsa_vec([topo.fsa] * B) if den else wf.compile_ctc_k2_graphs(topo, seqs)
    lp = torch.log_softmax(torch.randn(B, T, V, device=dev) * 3, -1).requires_grad_(True)
    seg = torch.tensor([[i, 0, T] for i in range(B)], dtype=torch.int32)
    out = {}
    for mode in ("exact", "pruned"):
        wf.set_output_beam_mode(mode)

        def step():
            lp.grad = None
            loss = wf.graphloss(g, wf.DenseFsaVec(lp, seg), output_beam=10, reduction="sum")
            loss.backward()
            return loss
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            l = step()
        torch.cuda.synchronize()
        out[mode] = ((time.perf_counter() - t0) / 20 * 1e3, float(l), lp.grad.clone())
    wf.set_output_beam_mode("exact")
    print("%-24s exact %.3f ms/step, pruned (beam 10) %.3f ms/step; loss %.4f vs %.4f, max |grad diff| %.2e" % (
        name, out["exact"][0], out["pruned"][0], out["exact"][1], out["pruned"][1],
        float((out["exact"][2] - out["pruned"][2]).abs().max())), flush=True)
