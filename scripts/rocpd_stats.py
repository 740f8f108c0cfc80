"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls, total / average duration, share.
usage: python scripts/rocpd_stats.py <results.db> [top_n]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
rows = db.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                  f"group by {name_col} order by sum(end-start) desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>8s} {'share':>6s}")
for n, c, s, a, mn, mx in rows[:top]:
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'\s*\[clone .*\]', '', n)
    print(f"{n[:70]:70s} {c:7d} {s/1e6:10.3f} {a/1e3:9.2f} {mn/1e3:8.2f} {mx/1e3:8.2f} {100*s/tot:5.1f}%")
