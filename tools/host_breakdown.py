"""Host-time breakdown of the drop-in step (cfg3): which Python / C pieces cost what (us per step)."""
import sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, ".")
import bench, waterfall_amd as wf
from waterfall_amd import _lib
from waterfall_amd import graph as G
dev = torch.device("cuda:0")
wl = bench.Workload("cfg3_ctc_bpe5k_num", 32, 500, dev, seed=777)
segs_t = torch.from_numpy(wl.segs)
x = wl.lp.detach().requires_grad_(True)
lib = _lib.lib()
acc = {}
def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r
    return w
for name in ("wf_fb_logsum", "wf_loss_reduce", "wf_grad_scale", "wf_fb_workspace_bytes"):
    setattr(lib, name, timed(name, getattr(lib, name)))
wf.DenseFsaVec.__init__ = timed("DenseFsaVec", wf.DenseFsaVec.__init__)
G._EagerLossFunction.forward = staticmethod(timed("eager.forward", G._EagerLossFunction.forward))
G._EagerLossFunction.backward = staticmethod(timed("eager.backward", G._EagerLossFunction.backward))
def step():
    x.grad = None
    dense = wf.DenseFsaVec(x, segs_t)
    loss = wf.graphloss(wl.fsa, dense, output_beam=10, reduction="sum")
    t0 = time.perf_counter(); loss.backward(); acc["loss.backward()"] = acc.get("loss.backward()", 0.0) + time.perf_counter() - t0
for _ in range(20): step()
torch.cuda.synchronize(); acc.clear()
N = 300
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter(); torch.cuda.synchronize()
print("host us/step total %.1f" % ((t1 - t0) / N * 1e6), {k: round(v / N * 1e6, 1) for k, v in acc.items()})
# the fused C call alone
acc.clear()
st = torch.cuda.current_stream(dev).cuda_stream
t0 = time.perf_counter()
for _ in range(N): wl.step(st)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("fused loop: host us/step %.1f wall %.1f" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6), {k: round(v / N * 1e6, 1) for k, v in acc.items()})
