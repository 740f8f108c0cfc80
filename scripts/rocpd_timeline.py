"""Timeline of ONE optimisation step out of a rocprofv3 (rocpd sqlite) kernel trace of the graph-replayed benchmark:
every kernel between two begin_step launches late in the trace, with start offset, duration and queue/stream id.
usage: python scripts/rocpd_timeline.py <results.db> [step_index_from_end]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
qcol = next((c for c in ('queue_id', 'stream_id', 'queue', 'stream') if c in cols), None)
sel = f"select {name_col}, start, end" + (f", {qcol}" if qcol else ", 0") + " from kernels order by start"
rows = db.execute(sel).fetchall()
marks = [i for i, r in enumerate(rows) if 'begin_step' in r[0]]
print(f"columns: {cols}")
if len(marks) < back + 2:
    back = len(marks) - 2
a, b = marks[-back - 1], marks[-back]
t0 = rows[a][1]
print(f"step of {b - a} kernels, span {(rows[b][1] - t0) / 1e3:.1f} us")
qs = sorted({r[3] for r in rows[a:b]})
busy = {q: 0 for q in qs}
for n, s, e, q in rows[a:b]:
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n)
    busy[q] += e - s
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{qs.index(q)}  {n[:90]}")
print({f"q{qs.index(q)}": round(v / 1e3, 1) for q, v in busy.items()})
