import sys; sys.path.insert(0, ".")
import numpy as np, torch, bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
stream = torch.cuda.current_stream(dev).cuda_stream
fused = torch.zeros(3, device=dev, dtype=torch.float64)
for dbl in (False, True):
    w = bench.Workload("cfg3_ctc_bpe5k_num", 32, 500, dev, seed=777, double=dbl)
    d = float(np.median(bench.time_steps(w, 20, 3, None, stream, fused, min_region_s=0.03)))
    print("double", dbl, "ms/step %.4f" % (d / 20 * 1e3), "fallbacks", w.fallbacks(stream), w.tot[:3].tolist())
