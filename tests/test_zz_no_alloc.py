"""wf_fb_logsum_static: the entry point that allocates nothing (include/wf_dwfst.h) against wf_fb_logsum.

Collected last (file name): a failure here must not hide the other GPU tests from `pytest -x`."""
import ctypes as C
import threading

import numpy as np
import pytest
import torch

from tests import util
from tests.test_gpu_parity import DEV, fused_logsum
from waterfall_amd import _lib

pytestmark = pytest.mark.gpu


def static_logsum(fsa, lp, segs, gscale=-1.0, init=True):
    """wf_fb_logsum_static through the C ABI; every device buffer exists before the call."""
    lib = _lib.lib()
    x = torch.from_numpy(np.ascontiguousarray(lp)).to(DEV)
    N, T, V = x.shape
    segs = np.ascontiguousarray(segs, dtype=np.int32)
    B = segs.shape[0]
    gh, gd = fsa.blob_host(), fsa.blob_dev(torch.device(DEV))
    tot = torch.empty(B, dtype=torch.float32, device=DEV)
    grad = torch.full_like(x, 7.0)   # every element must be overwritten
    gs = torch.full((B,), gscale, dtype=torch.float32, device=DEV)
    p = lambda t: C.c_void_p(t.data_ptr())
    npp = lambda a: C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)
    st = C.c_void_p(torch.cuda.current_stream(torch.device(DEV)).cuda_stream)
    with torch.cuda.device(DEV):
        if init:
            _lib.check(lib.wf_init())
        torch.cuda.synchronize()
        # "allocates nothing", three ways: torch's caching allocator saw no request; the device's free memory as the HIP
        # runtime reports it (hipMemGetInfo: sees any hipMalloc, the library's included) is unchanged; the library's own
        # counter of allocating calls (pinned ring, streams, events) did not move
        before = torch.cuda.memory_allocated(DEV)
        free_before = torch.cuda.mem_get_info(DEV)[0]
        n0, n1 = C.c_int64(-1), C.c_int64(-2)
        _lib.check(lib.wf_debug_alloc_count(C.byref(n0)))
        rc = lib.wf_fb_logsum_static(npp(gh), p(gd), npp(fsa.utt2fsa), p(x), x.stride(0), x.stride(1), N, T, V, npp(segs), B,
                                     0.0, 0, p(gs), p(tot), p(grad), st)
        _lib.check(lib.wf_debug_alloc_count(C.byref(n1)))
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated(DEV) == before
        assert torch.cuda.mem_get_info(DEV)[0] == free_before
        assert n1.value == n0.value
    if rc:
        return rc, lib.wf_last_error().decode()
    torch.cuda.synchronize()
    return tot.cpu().numpy(), grad.cpu().numpy()


def test_static_entry_point_matches_wf_fb_logsum_twice_in_a_row():
    # two calls in a row on the same GPU's scratch array, the second with a smaller input: stale contents of the
    # first call must not leak into the second
    big = util.random_case(901, "ctc", B=6, T=90, L=12, P=40)
    small = util.random_case(902, "2state", B=3, T=31, L=4, P=9)
    want = [fused_logsum(c["fsa"], c["lp"], c["segs"]) for c in (big, small)]
    got = [static_logsum(c["fsa"], c["lp"], c["segs"]) for c in (big, small)]
    for (wt, wg), (gt, gg) in zip(want, got):
        assert np.array_equal(wt, gt)          # same kernels, same arithmetic: bit-identical
        assert np.array_equal(wg, gg)
    # and a denominator (shared topology, hub state: another sweep kernel), then the first case again
    den = util.random_case(903, "mmictc", B=4, T=40, L=5, P=12, den=True)
    wt, wg = fused_logsum(den["fsa"], den["lp"], den["segs"])
    gt, gg = static_logsum(den["fsa"], den["lp"], den["segs"])
    assert np.array_equal(wt, gt) and np.array_equal(wg, gg)
    gt, gg = static_logsum(big["fsa"], big["lp"], big["segs"])
    assert np.array_equal(want[0][0], gt) and np.array_equal(want[0][1], gg)


def test_static_entry_point_rejects_what_does_not_fit_and_needs_init():
    lib = _lib.lib()
    n = C.c_size_t(0)
    _lib.check(lib.wf_static_scratch_bytes(C.byref(n)))
    assert n.value == 96 << 20
    # 8 utterances x 121 rows x 10 002 states x 8 B x (alpha + beta) = 155 MB > 96 MiB
    c = util.random_case(904, "mmictc", B=8, T=120, L=4, P=5000, den=True)
    rc, msg = static_logsum(c["fsa"], c["lp"], c["segs"])
    assert rc == -2 and "scratch" in msg       # WF_ENOSPC
    # a host thread that never called wf_init() has no context on the device
    small = util.random_case(905, "ctc", B=2, T=20, L=3, P=6)
    out = {}
    t = threading.Thread(target=lambda: out.update(r=static_logsum(small["fsa"], small["lp"], small["segs"], init=False)))
    t.start()
    t.join()
    assert out["r"][0] == -1 and "wf_init" in out["r"][1]   # WF_EINVAL
