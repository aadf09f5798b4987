"""Per-kernel register / scratch / LDS usage of wf_kernels.hip as hipcc reports it (no GPU needed).
usage: python tools/kernel_resources.py [substring ...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "waterfall_amd", "csrc", "wf_kernels.hip")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-o", "/tmp/wf_res.o", src,
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
want = sys.argv[1:]
for r in rows:
    if want and not any(w in r["name"] for w in want):
        continue
    print(f'{r["name"][:110]:110s} VGPR {r.get("VGPRs","?"):>4s} AGPR {r.get("AGPRs","?"):>4s} spill {r.get("VGPR Spill","?"):>3s} '
          f'scratch {r.get("ScratchSize [bytes/lane]","?"):>5s} occ {r.get("Occupancy [waves/SIMD]","?"):>2s} LDS {r.get("LDS Size [bytes/block]","?")}')
