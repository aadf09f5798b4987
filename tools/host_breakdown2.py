"""Line-level host-time breakdown of _EagerLossFunction.forward / backward pieces (us per call)."""
import sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, ".")
import bench, waterfall_amd as wf
from waterfall_amd import _lib
from waterfall_amd import graph as G
from waterfall_amd.graph import _check_inputs, _np_ptr, _ptr, _stream, _on, _const
dev = torch.device("cuda:0")
wl = bench.Workload("cfg3_ctc_bpe5k_num", 32, 500, dev, seed=777)
segs_t = torch.from_numpy(wl.segs)
x = wl.lp.detach().requires_grad_(True)
graph = wl.fsa
dense = wf.DenseFsaVec(x, segs_t)
N = 2000
def t(name, fn):
    for _ in range(50): fn()
    t0 = time.perf_counter()
    for _ in range(N): fn()
    print("%-40s %.2f us" % (name, (time.perf_counter() - t0) / N * 1e6))
lp = x
t("_check_inputs", lambda: _check_inputs(graph, dense))
t("lp.shape unpack", lambda: lp.shape)
t("blob_host+blob_dev", lambda: (graph.blob_host(), graph.blob_dev(dev)))
seg = dense.supervision_segments.numpy()
t("seg.numpy()", lambda: dense.supervision_segments.numpy())
t("workspace_bytes cached", lambda: graph.workspace_bytes(seg, 32, 5001, 0, True))
t("torch.empty ws 46MB", lambda: torch.empty(46231552, dtype=torch.uint8, device=dev))
t("torch.empty tot", lambda: torch.empty(32, dtype=torch.float32, device=dev))
t("torch.empty_like lp (320MB)", lambda: torch.empty_like(lp))
t("_const", lambda: _const(dev, 32, -1.0))
t("_stream", lambda: _stream(dev))
t("_on(dev) enter/exit", lambda: _on(dev).__enter__())
t("torch.empty(()) loss", lambda: torch.empty((), dtype=torch.float32, device=dev))
gh, gd = graph.blob_host(), graph.blob_dev(dev)
t("_np_ptr x3 + _ptr x6", lambda: (_np_ptr(gh), _np_ptr(None), _np_ptr(seg), _ptr(gd), _ptr(lp), _ptr(lp), _ptr(lp), _ptr(lp), _ptr(lp)))
t("lp.stride(0), stride(1)", lambda: (lp.stride(0), lp.stride(1)))
t("lp.is_contiguous()", lambda: lp.is_contiguous())
t("str(dev)", lambda: str(dev))
torch.cuda.synchronize()
