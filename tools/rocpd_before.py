"""what ran in the W ms before the n-th launch of a kernel (rocpd database), by kernel name and queue, plus the idle time.
usage: python tools/rocpd_before.py results.db kernel-substring [window_ms] [which]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, start, end, queue_id from kernels order by start").fetchall()
pat = sys.argv[2]; W = float(sys.argv[3]) if len(sys.argv) > 3 else 25.0
idx = [i for i, r in enumerate(rows) if pat in r[0]]
which = int(sys.argv[4]) if len(sys.argv) > 4 else -3
t1 = rows[idx[which]][1]; t0 = t1 - W * 1e6
seg = [r for r in rows if r[2] > t0 and r[1] < t1]
busy = sum(min(r[2], t1) - max(r[1], t0) for r in seg) / 1e6
print("window %.1f ms before launch %d of %s: %d kernels, sum of kernel time %.2f ms" % (W, which, pat, len(seg), busy))
agg = collections.OrderedDict()
for r in seg:
    k = (r[0].replace("hyp::", "")[:60], r[3])
    a = agg.setdefault(k, [0, 0.0, r[1]]); a[0] += 1; a[1] += (r[2] - r[1]) / 1e3
for (k, qid), (n, us, first) in agg.items():
    print("  q%-2s %5d x %-60s %9.1f us  first at -%.2f ms" % (qid, n, k, us, (t1 - first) / 1e6))
