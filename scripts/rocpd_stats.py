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

# busy fraction: union of the kernel intervals over the span of the trace (concurrent streams overlap)
iv = sorted(db.execute("select start, end from kernels").fetchall())
if iv:
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    busy += cur_e - cur_s
    span = iv[-1][1] - iv[0][0] if len(iv) > 1 else 1
    # the densest 60% of the trace (steady-state graph replays; the head is setup / capture)
    t0 = iv[0][0] + int(0.4 * (max(e for _, e in iv) - iv[0][0]))
    tail = [(a, b) for a, b in iv if a >= t0]
    tb, cs, ce = 0, tail[0][0], tail[0][1]
    for a, b in tail[1:]:
        if a > ce:
            tb += ce - cs
            cs, ce = a, b
        else:
            ce = max(ce, b)
    tb += ce - cs
    tspan = max(e for _, e in tail) - tail[0][0]
    print(f"trace span {span/1e6:.1f} ms, GPU busy (union of kernels) {busy/1e6:.1f} ms = {100*busy/span:.1f}%; "
          f"last 60% of the trace: busy {100*tb/tspan:.1f}% of {tspan/1e6:.1f} ms, sum of kernel durations {sum(b-a for a,b in tail)/1e6:.1f} ms")
