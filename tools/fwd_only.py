#!/usr/bin/env python3
"""Debug helper: forward-only (tot scores) sweep on the cfg3 workload, a few calls."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from waterfall_amd import _lib
dev = torch.device("cuda:0")
wl = bench.Workload(sys.argv[1] if len(sys.argv) > 1 else "cfg3_ctc_bpe5k_num", 32, 500, dev, 777)
lib = _lib.lib()
st = torch.cuda.current_stream(dev).cuda_stream
n = C.c_size_t(0)
_lib.check(lib.wf_fb_workspace_bytes(bench._np(wl.gh), bench._np(wl.fsa.utt2fsa), bench._np(wl.segs), wl.B, wl.B, wl.V, 0, 0, C.byref(n)))
wl.ws_bytes = max(wl.ws_bytes, n.value)
wl.ws = torch.empty(wl.ws_bytes, dtype=torch.uint8, device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for i in range(6):
    if i == 3:
        ev[0].record()
    lp = wl.lp
    _lib.check(lib.wf_fb_logsum(bench._np(wl.gh), bench._ptr(wl.gd), bench._np(wl.fsa.utt2fsa), bench._ptr(lp), lp.stride(0),
                                lp.stride(1), wl.B, wl.T, wl.V, bench._np(wl.segs), wl.B, 0.0, 0, None, bench._ptr(wl.tot),
                                None, bench._ptr(wl.ws), wl.ws_bytes, C.c_void_p(st)))
    torch.cuda.synchronize()
ev[1].record(); torch.cuda.synchronize()
print(wl.tot[:4], 'ms per forward-only call: %.3f' % (ev[0].elapsed_time(ev[1]) / 3))
