#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database into a per-kernel stats table (text).
usage: probes/rocpd_stats.py results.db [out.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                  f"from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
lines = [f"{'kernel':90s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}"]
for n, c, s, a, mn, mx in rows:
    lines.append(f"{n[:90]:90s} {c:6d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*s/tot:6.2f}")
lines.append(f"TOTAL kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
