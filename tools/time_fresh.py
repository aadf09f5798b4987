"""fresh-transcript step timings (bench.py dropin_times) for the headline workload only."""
import sys
sys.path.insert(0, ".")
import json
import torch
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wl = bench.Workload(sys.argv[1] if len(sys.argv) > 1 else "cfg3_ctc_bpe5k_num", 32, 500, dev, seed=777)
out = bench.dropin_times(wl, 50)
for k, v in out.items():
    print(k, json.dumps(v)[:200])
