import cProfile, pstats, torch, numpy as np, sys
sys.path.insert(0, '.')
import bench, waterfall_amd as wf
dev = torch.device("cuda:0")
wl = bench.Workload("cfg3_ctc_bpe5k_num", 32, 500, dev, seed=777)
segs_t = torch.from_numpy(wl.segs)
x = wl.lp.detach().requires_grad_(True)
def dropin():
    x.grad = None
    dense = wf.DenseFsaVec(x, segs_t)
    loss = wf.graphloss(wl.fsa, dense, output_beam=10, reduction="sum")
    loss.backward()
for _ in range(20): dropin()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): dropin()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
