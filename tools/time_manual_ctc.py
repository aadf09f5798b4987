"""bench.py's manual_ctc line alone."""
import sys
sys.path.insert(0, ".")
import json
import torch
import bench
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
print(json.dumps(bench.manual_ctc_bench(dev), indent=1))
print(json.dumps(bench.manual_ctc_bench(dev, V=5001, L=64), indent=1))
