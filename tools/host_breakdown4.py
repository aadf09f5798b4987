"""Statement-level timing of the eager backward inside the real loop (us, averaged)."""
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
acc = np.zeros(12); cnt = [0]
P = time.perf_counter
def backward(ctx, gout):
    t = [P()]
    grad, ctx.grad = ctx.grad, None
    graph, dense = ctx.graph, ctx.dense
    seg = dense.supervision_segments.numpy()
    N, T, V = ctx.shape; t.append(P())
    (lp,) = ctx.saved_tensors; t.append(P())
    dev = lp.device
    gs = gout if (gout.dtype == torch.float32 and gout.is_contiguous()) else gout.to(torch.float32).contiguous(); t.append(P())
    c = _on(dev); t.append(P())
    st = _stream(dev); t.append(P())
    a = (_np_ptr(graph.blob_host()), _ptr(graph.blob_dev(dev)), _np_ptr(graph.utt2fsa),
                                                _np_ptr(seg), seg.shape[0], N, T, V, ctx.strides[0], ctx.strides[1], _ptr(gs),
                                                1 if ctx.per_utt else 0, _ptr(grad), _ptr(ctx.ws), ctx.ws_bytes, st); t.append(P())
    _lib.check(_lib.lib().wf_grad_scale(*a)); t.append(P())
    d = np.diff(t); acc[:d.size] += d; cnt[0] += 1
    return grad, None, None, None, None
G._EagerLossFunction.backward = staticmethod(backward)
tb = [0.0]
def step():
    x.grad = None
    dense = wf.DenseFsaVec(x, segs_t)
    loss = wf.graphloss(wl.fsa, dense, output_beam=10, reduction="sum")
    t0 = P(); loss.backward(); tb[0] += P() - t0
for _ in range(20): step()
torch.cuda.synchronize(); acc[:] = 0; cnt[0] = 0; tb[0] = 0
for _ in range(300): step()
torch.cuda.synchronize()
names = ["ctx fields+seg.numpy", "saved_tensors", "gs", "_on", "_stream", "args", "wf_grad_scale"]
print({n: round(a / cnt[0] * 1e6, 1) for n, a in zip(names, acc)}, "loss.backward() total %.1f" % (tb[0] / 300 * 1e6))
