import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import waterfall_amd as wf
P, L, B = 5000, 64, 32
topo = wf.Topology(sys.argv[1] if len(sys.argv) > 1 else "ctc", ["<eps>", "<SIL>"] + ["p%d" % i for i in range(P)])
rng = np.random.default_rng(0)
for _ in range(3):
    f = wf.compile_ctc_k2_graphs(topo, (rng.integers(2, P + 2, size=B * L).astype(np.int32), np.full(B, L)), device="cuda:0")
    f.blob_host()
torch.cuda.synchronize()
