"""Where the host's time goes in the fresh-graphs step (device compiler one batch ahead + fused wf_fb_logsum): cProfile
over the same loop as bench.py's `fresh_graphs` line."""
import sys
sys.path.insert(0, ".")
import cProfile
import ctypes as C
import pstats
import time
import numpy as np
import torch
import bench
import waterfall_amd as wf
from waterfall_amd import _lib
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wl = bench.Workload("cfg3_ctc_bpe5k_num", 32, 500, dev, seed=777)
w = bench.WORKLOADS[wl.name]
wf.set_output_beam_mode("exact")
rng = np.random.default_rng(4242)
P, L, B = w["P"], w["L"], wl.B
topo = wf.Topology(w["topo"], ["<eps>", "<SIL>"] + ["p%d" % i for i in range(P)])
batches = [(rng.integers(2, P + 2, size=B * L).astype(np.int32), np.full(B, L)) for _ in range(53)]
it = [0]
lib = _lib.lib()
stream = torch.cuda.current_stream(dev).cuda_stream
_np, _ptr = bench._np, bench._ptr


def compile_dev():
    b = batches[it[0] % len(batches)]
    it[0] += 1
    return wf.compile_ctc_k2_graphs(topo, b, device=dev)


def fused_on(fsa):
    gh, gd = fsa.blob_host(), fsa.blob_dev(dev)
    lp = wl.lp
    n = C.c_size_t(0)
    _lib.check(lib.wf_fb_workspace_bytes(_np(gh), None, _np(wl.segs), wl.B, wl.B, wl.V, 0, 1, C.byref(n)))
    ws = torch.empty(n.value, dtype=torch.uint8, device=dev)
    _lib.check(lib.wf_fb_logsum(_np(gh), _ptr(gd), None, _ptr(lp), lp.stride(0), lp.stride(1), wl.B, wl.T, wl.V,
                                _np(wl.segs), wl.B, 0.0, 0, _ptr(wl.gscale), _ptr(wl.tot), _ptr(wl.grad), _ptr(ws),
                                n.value, C.c_void_p(stream)))


pending = [compile_dev()]


def step():
    nxt = compile_dev()
    fused_on(pending.pop())
    pending.append(nxt)


x = wl.lp.detach().requires_grad_(True)
segs_t = torch.from_numpy(wl.segs)


def step_dropin():
    nxt = compile_dev()
    x.grad = None
    dense = wf.DenseFsaVec(x, segs_t)
    loss = wf.graphloss(pending.pop(), dense, output_beam=10, reduction="sum")
    loss.backward()
    pending.append(nxt)


def step_resident():
    x.grad = None
    dense = wf.DenseFsaVec(x, segs_t)
    loss = wf.graphloss(wl.fsa, dense, output_beam=10, reduction="sum")
    loss.backward()


mode = sys.argv[1] if len(sys.argv) > 1 else "fused"
fn = {"dropin": step_dropin, "resident": step_resident, "fused": step}[mode]
for _ in range(50):
    fn()
torch.cuda.synchronize()
N = 2000
t0 = time.perf_counter()
for _ in range(N):
    fn()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host issue %.1f us/step, wall %.1f us/step" % ((t1 - t0) / N * 1e6, (time.perf_counter() - t0) / N * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    fn()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
rows = []
for (f, ln, name), (cc, nc, tt, ct, _) in st.stats.items():
    rows.append((tt / N * 1e6, ct / N * 1e6, nc / N, "%s:%d %s" % (f.split("/")[-1], ln, name)))
rows.sort(reverse=True)
print("%8s %8s %6s  function  (us per step: own, cumulative; calls per step)" % ("own", "cum", "calls"))
for r in rows[:32]:
    print("%8.2f %8.2f %6.1f  %s" % r)
