#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
export HYP_POTRF_SPLIT=0
echo "--- default"; python tools/bench_potrf.py 5000
for w in 1 4; do echo "--- panel waves $w"; HYP_PANEL_WAVES=$w python tools/bench_potrf.py 5000; done
for m in 768 1024 2048 2560; do echo "--- la_min $m"; HYP_POTRF_LA_MIN=$m python tools/bench_potrf.py 5000; done
echo "--- own cu 0"; HYP_POTRF_OWN_CU=0 python tools/bench_potrf.py 5000
echo "--- trail tile 0"; HYP_POTRF_TRAIL_TILE=0 python tools/bench_potrf.py 5000
echo "--- look tile 64"; HYP_POTRF_LOOK_TILE=64 python tools/bench_potrf.py 5000
rm -rf /tmp/prof_p; rocprofv3 --kernel-trace -d /tmp/prof_p -o b -- python tools/bench_potrf.py 5000 > /dev/null 2>&1
DB=$(find /tmp/prof_p -name "*.db" | head -1)
python - $DB > gpurun_out/cholesky_timeline_tiles4.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = cur.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()[-196:]
t0 = rows[0][1]
for r in rows:
    print("%10.1f us  dur %8.1f us  q%-3s %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[0].replace("hyp::", "")[:60]))
PY
sed -n 1,40p gpurun_out/cholesky_timeline_tiles4.txt
