"""Host time of the two C calls of the autograd path (cfg3)."""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
import waterfall_amd as wf
from waterfall_amd import _lib
dev = torch.device("cuda:0")
wl = bench.Workload(sys.argv[1] if len(sys.argv) > 1 else "cfg3_ctc_bpe5k_num", 32, 500, dev, 777)
segs = torch.from_numpy(wl.segs)
x = wl.lp.clone().requires_grad_(True)
lib = _lib.lib()
acc = {"wf_fb_logsum": 0.0, "wf_fb_backward": 0.0, "wf_fb_workspace_bytes": 0.0}
for name in acc:
    fn = getattr(lib, name)
    def wrap(*a, _fn=fn, _name=name):
        t0 = time.perf_counter()
        r = _fn(*a)
        acc[_name] += time.perf_counter() - t0
        return r
    setattr(lib, name, wrap)
def step():
    x.grad = None
    dense = wf.DenseFsaVec(x, segs)
    loss = wf.graphloss(wl.fsa, dense, output_beam=10, reduction="sum")
    loss.backward()
for _ in range(10): step()
torch.cuda.synchronize()
for k in acc: acc[k] = 0.0
t0 = time.perf_counter()
for _ in range(100): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host us/step: total %.1f" % ((t1 - t0) * 1e4), {k: round(v * 1e4, 1) for k, v in acc.items()})
