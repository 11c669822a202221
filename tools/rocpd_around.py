"""Kernel rows of a rocprofv3 rocpd trace around the n-th LAST launch of a marker kernel.
usage: python tools/rocpd_around.py results.db marker-substring [nth_from_last=1] [before_ms=2] [after_ms=2]
-> 'start_us(relative to the marker) dur_us queue name'"""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
marker = sys.argv[2]
nth = int(sys.argv[3]) if len(sys.argv) > 3 else 1
before = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
after = float(sys.argv[5]) if len(sys.argv) > 5 else 2.0
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = cur.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()
hits = [i for i, r in enumerate(rows) if marker in r[0]]
t0 = rows[hits[-nth]][1]
for name, s, e, qd in rows:
    if s < t0 - before * 1e6 or s > t0 + after * 1e6:
        continue
    print("%10.1f %8.1f q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, qd, name.replace("hyp::", "").replace("(anonymous namespace)::", "")[:70]))
