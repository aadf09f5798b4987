#!/usr/bin/env python3
"""Kernel timeline of the last step in a rocprofv3 rocpd database: start / end (us, relative) and name of the last N
kernel dispatches.  python tools/prof_timeline.py x_results.db [N [name]]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else None
if view is None:
    print("tables:", tabs)
    sys.exit(1)
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
rows = list(db.execute(f"select name, start, end, stream_id from {view} order by start")) if "stream_id" in cols else \
    [r + (0,) for r in db.execute(f"select name, start, end from {view} order by start")]
pat = sys.argv[3] if len(sys.argv) > 3 else None
if pat:     # a window of n dispatches that starts at the 20th dispatch whose name contains pat
    hits = [i for i, r in enumerate(rows) if pat in r[0]]
    i0 = hits[min(20, len(hits) - 1)] if hits else 0
    rows = rows[i0:i0 + n]
else:
    rows = rows[-n:]
t0 = rows[0][1]
for name, s, e, sid in rows:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  stream {sid}  {name[:90]}")
