"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd sqlite db.  usage: pmc_summary.py <db> [--schema]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
if '--schema' in sys.argv:
    for t in tabs:
        if re.search(r'_[0-9a-f]{8}_', t):
            continue
        cols = [r[1] for r in db.execute(f'pragma table_info({t})')]
        print(t, cols)
    sys.exit(0)
# rocprofv3 ships a 'counters_collection' view: one row per (dispatch, counter)
view = 'counters_collection' if 'counters_collection' in tabs else None
if view is None:
    print('no counters_collection view; tables:', [t for t in tabs if not re.search(r'_[0-9a-f]{8}_', t)])
    sys.exit(1)
cols = [r[1] for r in db.execute(f'pragma table_info({view})')]
kcol = 'kernel_name' if 'kernel_name' in cols else [c for c in cols if 'kernel' in c and 'name' in c][0]
ccol = 'counter_name' if 'counter_name' in cols else [c for c in cols if 'counter' in c and 'name' in c][0]
vcol = 'value' if 'value' in cols else [c for c in cols if 'value' in c][0]
rows = db.execute(f"select {kcol}, {ccol}, count(*), sum({vcol}), avg({vcol}) from {view} group by {kcol}, {ccol} order by sum({vcol}) desc").fetchall()
print(f"{'kernel':60s} {'counter':28s} {'n':>7s} {'sum':>16s} {'avg':>14s}")
for k, c, n, s, a in rows[:60]:
    k = re.sub(r'\(anonymous namespace\)::', '', k)[:60]
    print(f"{k:60s} {c:28s} {n:7d} {s:16.1f} {a:14.2f}")
