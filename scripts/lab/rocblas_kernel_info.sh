#!/bin/bash
# Which kernel does rocBLAS pick for the update's carrying NT shape, and with what resources (rocprofv3 kernel trace of
# scripts/lab/blas_ref): the Tensile kernel name encodes macro-tile, MFMA instruction, LDS / prefetch options.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for s in "16384 1024 1024" "131072 1024 1024"; do
  rm -rf /tmp/rb
  rocprofv3 --kernel-trace --stats -d /tmp/rb -o t -- scripts/lab/blas_ref $s 20 > /dev/null 2>&1
  echo "== $s"
  python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob('/tmp/rb/**/t_results.db', recursive=True)[0])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, count(*), avg(end-start)/1e3, max(workgroup_x), max(grid_x), max(lds_size), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(scratch_size) from kernels group by name order by 3 desc").fetchall()
for r in rows[:4]:
    print(r)
PY
done
