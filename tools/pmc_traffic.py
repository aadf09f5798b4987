#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs, as
MI355X_MICROARCH.md prescribes: they do not fit one pass).  Units: the counters are KB; on gfx950 FETCH_SIZE counts
64-byte requests as 32 B for wide reads, so the guide's correction doubles it (both figures are kept).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc -o fetch -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc -o write -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc/fetch_results.db gpurun_out/pmc/write_results.db "<workload>" > profiles/rNN_pmc_traffic.json
"""
import json
import re
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    out = {}
    for name, val in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        m = re.search(r"(\w+_kernel)\b", name)
        if not m or "anonymous namespace)::" not in name or "at::native" in name:
            continue
        out.setdefault(m.group(1), []).append(float(val))
    # MEDIAN launch, not the mean: one kernel name covers launches that do different work (sweep_band_kernel with and
    # without the fill workgroups -- bench.py's per-kernel timing leg sends the fill to the side stream for 20 launches;
    # round 2's mean mixed 114 launches of 353 MB with 20 of 33 MB into "306 MB")
    med = lambda v: sorted(v)[len(v) // 2]
    return {k: med(v) for k, v in out.items()}, {k: len(v) for k, v in out.items()}


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
res = {"workload": sys.argv[3] if len(sys.argv) > 3 else "",
       "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --kernel-trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline; tools/pmc_traffic.py",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    res["kernels"][k] = {"launches": nf.get(k, 0), "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
                         "hbm_bytes_per_launch_raw": (f + w) * 1024, "hbm_bytes_per_launch_fetch_doubled": (2 * f + w) * 1024}
print(json.dumps(res, indent=1))
