"""Print the kernel timeline of a rocprofv3 rocpd trace after the LAST launch of a marker kernel.
usage: python tools/rocpd_timeline.py results.db [marker-substring] [max-rows] [rows-before-the-marker]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "gemm_f64_kernel<true, 4, 1>"
nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 120
back = int(sys.argv[4]) if len(sys.argv) > 4 else 0
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
sel = "name, start, end" + (", " + qcol if qcol else "")
rows = cur.execute("select %s from kernels order by start" % sel).fetchall()
last = max((i for i, r in enumerate(rows) if marker in r[0]), default=0)
t0 = rows[last][1]
for r in rows[max(0, last - back):last + nmax]:
    name = r[0].replace("hyp::", "")
    name = name[:60]
    print("%10.1f us  dur %8.1f us  q%-3s %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3] if qcol else "", name))
