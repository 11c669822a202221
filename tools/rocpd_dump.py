"""Raw kernel rows of a rocprofv3 rocpd trace for offline analysis: python tools/rocpd_dump.py results.db [window_ms] [skip_ms]
-> 'start_us dur_us queue name' of the kernels in the window of window_ms that ends skip_ms before the last kernel of the trace."""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
win = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
skip = float(sys.argv[3]) if len(sys.argv) > 3 else 50.0
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = cur.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()
t_end = rows[-1][2] - skip * 1e6
t_beg = t_end - win * 1e6
for r in rows:
    if t_beg <= r[1] <= t_end:
        print("%.1f %.1f %s %s" % ((r[1] - t_beg) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[0].replace("hyp::", "").replace("(anonymous namespace)::", "")[:90]))
