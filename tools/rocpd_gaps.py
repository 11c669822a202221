"""GPU busy / idle analysis of one IPM iteration (between the last two Schur syrk launches) of a rocprofv3
kernel trace.  usage: python tools/rocpd_gaps.py results.db [from_us to_us]  (optional window: print the timeline)"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.cursor().execute("select name,start,end from kernels order by start").fetchall()
syrk = [i for i, r in enumerate(rows) if "gemm_f64_kernel<true, 4, 1>" in r[0]]
import os
back = int(os.environ.get("ITER_BACK", "1"))   # 1 = the last iteration of the trace, 2 = the one before, ...
a, b = syrk[-1 - back], syrk[-back]
t0, t1 = rows[a][1], rows[b][1]
print("iteration window %.2f ms, %d kernels" % ((t1 - t0) / 1e6, b - a))
busy, last_end, gaps, bykern = 0, t0, [], collections.Counter()
for r in rows[a:b]:
    st, en = r[1], r[2]
    if st > last_end:
        gaps.append((st - last_end, (last_end - t0) / 1e3, r[0][:50]))
    busy += max(0, en - max(st, last_end))
    last_end = max(last_end, en)
    bykern[r[0][:70].replace("hyp::", "")] += en - st
print("busy %.2f ms, idle %.2f ms" % (busy / 1e6, (t1 - t0 - busy) / 1e6))
print("idle in gaps >20us: %.2f ms (%d); 5-20us: %.2f ms (%d); <5us: %.2f ms (%d)" % (
    sum(g[0] for g in gaps if g[0] > 2e4) / 1e6, sum(g[0] > 2e4 for g in gaps),
    sum(g[0] for g in gaps if 5e3 < g[0] <= 2e4) / 1e6, sum(5e3 < g[0] <= 2e4 for g in gaps),
    sum(g[0] for g in gaps if g[0] <= 5e3) / 1e6, sum(g[0] <= 5e3 for g in gaps)))
for g in sorted(gaps, reverse=True)[:12]:
    print("  gap %8.1f us at %9.1f us before %s" % (g[0] / 1e3, g[1], g[2]))
for k, v in bykern.most_common(16):
    print("  %8.3f ms  %s" % (v / 1e6, k))
if len(sys.argv) > 3:
    lo, hi = float(sys.argv[2]), float(sys.argv[3])
    prev = None
    for r in rows[a:b]:
        rel = (r[1] - t0) / 1e3
        if lo < rel < hi:
            print("%9.1f gap %7.1f dur %7.1f %s" % (rel, (r[1] - prev) / 1e3 if prev else 0, (r[2] - r[1]) / 1e3, r[0][:70].replace("hyp::", "")))
        prev = r[2]
