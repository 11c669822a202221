"""Condensed kernel timeline of a window of a rocprofv3 rocpd trace: runs of equal kernel names with their count, summed
duration and summed gaps.  usage: python tools/rocpd_window.py results.db from_ms_before_end length_ms"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back, length = float(sys.argv[2]), float(sys.argv[3])
rows = db.cursor().execute("select name,start,end from kernels order by start").fetchall()
tend = rows[-1][2]
t0 = tend - back * 1e6
t1 = t0 + length * 1e6
prev, cnt, tot, gaps, st, last, pend = None, 0, 0.0, 0.0, 0.0, 0.0, None
for name, s, e in rows:
    if s < t0 or s > t1:
        continue
    nm = name.replace("hyp::", "").replace("(anonymous namespace)::", "")[:56]
    gap = max(0.0, (s - pend) / 1e3) if pend is not None else 0.0
    if nm == prev:
        cnt += 1; tot += (e - s) / 1e3; gaps += gap; last = (e - t0) / 1e3
    else:
        if prev is not None:
            print("%9.1f..%9.1f  x%-4d dur %8.1f gaps %7.1f  %s" % (st, last, cnt, tot, gaps, prev))
        prev, cnt, tot, gaps, st, last = nm, 1, (e - s) / 1e3, gap, (s - t0) / 1e3, (e - t0) / 1e3
    pend = max(pend, e) if pend is not None else e
if prev is not None:
    print("%9.1f..%9.1f  x%-4d dur %8.1f gaps %7.1f  %s" % (st, last, cnt, tot, gaps, prev))
