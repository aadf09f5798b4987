#!/usr/bin/env python3
"""Dump the kernel summary (the --stats view) of a rocprofv3 rocpd database as text.

    python tools/prof_summary.py gpurun_out/prof_x/x_results.db > profiles/x_kernel_stats.txt
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel")
for name, calls, tot, avg, pct in rows:
    if len(name) > 150:
        name = name[:147] + "..."
    print(f"{calls:>6} {tot:>12.3f} {avg:>10.3f} {pct:>6.2f}  {name}")
