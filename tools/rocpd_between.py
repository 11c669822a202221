"""kernels between the last launch of kernel A and the next launch of kernel B (rocpd database): what is still queued there.
usage: python tools/rocpd_between.py results.db A-substring B-substring"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, start, end, queue_id from kernels order by start").fetchall()
A, B = sys.argv[2], sys.argv[3]
ia = [i for i, r in enumerate(rows) if A in r[0]]
for pick in ia[-40::13]:
    nxt = next((j for j in range(pick + 1, len(rows)) if B in rows[j][0]), None)
    if nxt is None: continue
    seg = rows[pick:nxt + 1]
    print("---- %d kernels, %.2f ms from the end of A to the start of B" % (len(seg) - 2, (seg[-1][1] - seg[0][2]) / 1e6))
    agg = collections.OrderedDict()
    for r in seg[1:-1]:
        k = r[0].replace("hyp::", "")[:70]
        a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (r[2] - r[1]) / 1e3
    for k, (n, us) in agg.items():
        print("   %5d x %-70s %9.1f us" % (n, k, us))
