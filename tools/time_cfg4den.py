"""cfg4's two hub-form denominators: ms per fused loss + gradient step (bench.py's lines alone)."""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
stream = torch.cuda.current_stream(dev).cuda_stream
fused = torch.zeros(3, device=dev, dtype=torch.float64)
for name in ("cfg4_mmictc_bpe5k_den", "cfg4_3state-skip_bpe5k_den"):
    w = bench.Workload(name, 32, 500, dev, seed=777)
    d = float(np.median(bench.time_steps(w, 10, 3, None, stream, fused, min_region_s=0.05)))
    print(name, "ms/step %.4f" % (d / 10 * 1e3), "fallbacks", w.fallbacks(stream), flush=True)
    del w
    torch.cuda.empty_cache()
