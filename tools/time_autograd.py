"""Wall time per training step of the drop-in path: wf.graphloss(...) + backward() through torch autograd (cfg3)."""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
import waterfall_amd as wf
dev = torch.device("cuda:0")
wl = bench.Workload(sys.argv[1] if len(sys.argv) > 1 else "cfg3_ctc_bpe5k_num", 32, 500, dev, 777)
segs = torch.from_numpy(wl.segs)
x = wl.lp.clone().requires_grad_(True)
def step():
    x.grad = None
    dense = wf.DenseFsaVec(x, segs)
    loss = wf.graphloss(wl.fsa, dense, output_beam=10, reduction="sum")
    loss.backward()
for _ in range(10):
    step()
torch.cuda.synchronize()
n = 100
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("autograd path: host %.1f us/step issue, %.1f us/step total (fused C call in bench.py: see ms_per_step)" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
