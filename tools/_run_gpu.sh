python -m pytest tests/test_hip_dense.py -q -x -m gpu 2>&1 | tail -2
python tools/bench_potrf.py 5000 4845 2250 1000
HYP_POTRF_SPLIT3=0 python tools/bench_potrf.py 5000 2250
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d gpurun_out/prof_potrf3 -o p -- python tools/bench_potrf.py 5000 > /dev/null 2>&1
DB=$(find gpurun_out/prof_potrf3 -name "*.db" | head -1)
python - "$DB" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = cur.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()
# last factorization: find the last copyBuffer and print the 60 kernels after it from its middle
idx = [i for i, r in enumerate(rows) if "potrf_diag_mfma" in r[0]]
start = idx[-30]
t0 = rows[start][1]
out = open("gpurun_out/r02_cholesky_timeline.txt", "w")
for r in rows[start:start + 70]:
    out.write("%10.1f us  dur %8.1f us  q%-3s %s\n" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[0].replace("hyp::", "")[:70]))
out.close()
PY
head -40 gpurun_out/r02_cholesky_timeline.txt; rm -rf gpurun_out/prof_potrf3
