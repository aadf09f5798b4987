#!/usr/bin/env python3
"""Pretty-print the JSON line of a bench.py log."""
import json
import sys
for line in open(sys.argv[1]):
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    print(f"value {d['value']:.4g} {d['unit']}  ms/step {d['ms_per_step']:.4f}  n_gpus {d['n_gpus']}")
    for k, v in d.get("kernels", {}).items():
        print(f"   {k:7s} {v['avg_ms']*1e3:9.1f} us  {v['GBps'] or 0:8.1f} GB/s")
    if "roofline" in d:
        r = d["roofline"]
        print(f"   roofline: {r['kernel']} {r['achieved']:.1f}/{r['peak']} GB/s frac {r['frac']:.4f}")
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        print(f"   cpu: {c['value']:.4g} frames/s on {c['cores']} cores ({c['kind']})")
    for k, v in d.get("other_workloads", {}).items():
        if "error" in v:
            print(f"   {k:28s} ERROR {v['error']}")
            continue
        if "num_den_ms_per_step" in v:
            print(f"   {k:28s} two calls {v['two_call_ms_per_step']:.3f} ms/step, graphloss_num_den {v['num_den_ms_per_step']:.3f} ms/step")
            continue
        if "speedup_vs_exact" in v:
            print(f"   {k:28s} {v['ms_per_call']:.3f} ms/call ({v['ms_per_call_with_log_total']:.3f} with log total)  exact sweep {v['exact_sweep_ms_per_call']:.3f} ms  x{v['speedup_vs_exact']:.2f}  same path {v['same_path']}")
            continue
        if "ms_per_call" in v:
            print(f"   {k:28s} {v['frames_per_s']:.4g} f/s  {v['ms_per_call']:.3f} ms/call  {v['us_per_frame']:.1f} us/frame  {v['launch_model_GBps']:.0f} GB/s (bytes a launch must move)")
            continue
        if "ratio_to_exact" in v:
            print(f"   {k:28s} {v['frames_per_s']:.4g} f/s  {v['ms_per_step']:.4f} ms  (exact {v['exact_ms_per_step']:.4f} ms, x{v['ratio_to_exact']:.2f}; fallbacks {v['fallback_utterances']})")
            continue
        print(f"   {k:28s} {v['frames_per_s']:.4g} f/s  {v['ms_per_step']:.4f} ms  " +
              " ".join(f"{a}={b*1e3:.1f}us" for a, b in v.get("kernels_ms", {}).items() if not isinstance(b, bool)) +
              (f" graph build {v['graph_build_ms']:.3f} ms" if "graph_build_ms" in v else "") +
              (f" (float path {v['float_ms_per_step']:.4f} ms)" if "float_ms_per_step" in v else ""))
