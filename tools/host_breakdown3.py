"""Statement-level timing of the eager forward inside the real loop (us, averaged)."""
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
def forward(ctx, log_probs, graph, dense, reduction, weights):
    t = [P()]
    lp, seg, B = _check_inputs(graph, dense)
    lib = _lib.lib()
    dev = lp.device
    N, T, V = lp.shape
    gh, gd = graph.blob_host(), graph.blob_dev(dev)
    nbytes = graph.workspace_bytes(seg, N, V, 0, True); t.append(P())
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev); t.append(P())
    tot = torch.empty(B, dtype=torch.float32, device=dev); t.append(P())
    grad = torch.empty_like(lp); t.append(P())
    gscale = weights if weights is not None else _const(dev, B, -1.0)
    st = _stream(dev); t.append(P())
    with _on(dev):
        _lib.check(lib.wf_fb_logsum(_np_ptr(gh), _ptr(gd), _np_ptr(graph.utt2fsa), _ptr(lp), lp.stride(0), lp.stride(1),
                                    N, T, V, _np_ptr(seg), B, 0.0, 0, _ptr(gscale), _ptr(tot), _ptr(grad), _ptr(ws),
                                    nbytes, st)); t.append(P())
        loss = torch.empty((), dtype=torch.float32, device=dev); t.append(P())
        _lib.check(lib.wf_loss_reduce(_ptr(tot), _ptr(None), B, _ptr(loss), st)); t.append(P())
    ctx.graph, ctx.dense, ctx.ws, ctx.ws_bytes, ctx.grad, ctx.per_utt = graph, dense, ws, nbytes, grad, reduction == 'none'
    ctx.shape = (N, T, V)
    ctx.strides = (lp.stride(0), lp.stride(1))
    ctx.gscale = gscale
    ctx.save_for_backward(lp); t.append(P())
    d = np.diff(t); acc[:d.size] += d; cnt[0] += 1
    return loss
G._EagerLossFunction.forward = staticmethod(forward)
def step():
    x.grad = None
    dense = wf.DenseFsaVec(x, segs_t)
    loss = wf.graphloss(wl.fsa, dense, output_beam=10, reduction="sum")
    loss.backward()
for _ in range(20): step()
torch.cuda.synchronize(); acc[:] = 0; cnt[0] = 0
for _ in range(300): step()
torch.cuda.synchronize()
names = ["check+ws_bytes", "empty ws", "empty tot", "empty_like grad", "const+stream", "wf_fb_logsum", "empty loss", "wf_loss_reduce", "ctx+save"]
print({n: round(a / cnt[0] * 1e6, 1) for n, a in zip(names, acc)})
