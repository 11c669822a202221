"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max.
usage: python tools/rocpd_stats.py results.db [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
lines = ["name,calls,total_ms,avg_us,min_us,max_us,pct"]
for name, n, s, a, mn, mx in rows:
    short = name if len(name) < 90 else name[:87] + "..."
    lines.append('"%s",%d,%.3f,%.2f,%.2f,%.2f,%.2f' % (short, n, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
txt = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
print(txt)
