export TMPDIR=/tmp
cd /root/repo; timeout 300 rocprofv3 --kernel-trace -d /tmp/pp -o b -- python tools/bench_potrf.py 5000 > /tmp/pp.log 2>&1; tail -2 /tmp/pp.log
cd /root/repo; DB=$(find /tmp/pp -name "*.db" | head -1)
python - <<PY
import sqlite3
db=sqlite3.connect("$DB"); cur=db.cursor()
cols=[r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol="queue_id" if "queue_id" in cols else "stream_id"
rows=cur.execute("select name,start,end,%s from kernels order by start" % qcol).fetchall()
# last factorization: find last index of the fillBuffer preceding fused kernels
idx=[i for i,r in enumerate(rows) if "potrf_step_fused" in r[0]]
# start of last factorization = the fused kernel whose predecessor fused kernel is > 39 back ... take last 39 fused
start=idx[-39]
t0=rows[start][1]
for r in rows[start:start+60]:
    print("%9.1f us dur %7.1f q%s %s" % ((r[1]-t0)/1e3,(r[2]-r[1])/1e3,r[3],r[0].replace("hyp::","")[:50]))
print("...")
for r in rows[idx[-8]:idx[-8]+24]:
    print("%9.1f us dur %7.1f q%s %s" % ((r[1]-t0)/1e3,(r[2]-r[1])/1e3,r[3],r[0].replace("hyp::","")[:50]))
PY
