#!/usr/bin/env python3
"""Summarise a rocprofv3 results database (rocpd sqlite) into a per-kernel table:
   python tools/rocpd_summary.py gpurun_out/prof/x_results.db [> profiles/name.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
tot = sum(r[2] for r in rows) or 1
print("%-90s %8s %12s %12s %12s %12s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
for n, c, s, a, mn, mx in rows:
    short = n if len(n) <= 90 else n[:87] + "..."
    print("%-90s %8d %12.3f %12.2f %12.2f %12.2f %6.2f" % (short, c, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
print("total kernel time %.3f ms" % (tot / 1e6))
