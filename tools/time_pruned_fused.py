"""Fused wf_fb_logsum call with output_beam pruning (WF_PRUNED) against the exact call, per step, through bench.py's
Workload: cfg3 numerators, cfg2 numerators / denominators, cfg4 numerators.  usage: python tools/time_pruned_fused.py [beam]"""
import sys
import time
sys.path.insert(0, ".")
import torch
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
stream = torch.cuda.current_stream(dev).cuda_stream
beam = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
for name in ("cfg3_ctc_bpe5k_num", "cfg2_2state_char_num", "cfg2_2state_char_den", "cfg4_mmictc_bpe5k_num", "cfg4_3state-skip_bpe5k_num"):
    res = {}
    for b in (0.0, beam):
        wl = bench.Workload(name, 32, 500, dev, seed=777, beam=b)
        for _ in range(5):
            wl.step(stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            wl.step(stream)
        torch.cuda.synchronize()
        res[b] = ((time.perf_counter() - t0) / 50 * 1e3, wl.fallbacks(stream))
        del wl
        torch.cuda.empty_cache()
    print("%-28s exact %.4f ms/step (fallbacks %d)   beam %g: %.4f ms/step (fallbacks %d)   ratio %.2f" % (
        name, res[0.0][0], res[0.0][1], beam, res[beam][0], res[beam][1], res[beam][0] / res[0.0][0]), flush=True)
