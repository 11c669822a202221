export TMPDIR=/tmp
for few in 0 8 64; do
rm -rf /tmp/p$few; HYP_TS_FEW=$few rocprofv3 --kernel-trace -d /tmp/p$few -o b -- python bench.py --steps 30 --cpu-iters 0 > /dev/null 2>&1
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob('/tmp/p$few/**/*.db', recursive=True)[0])
for nm in ('psd_ts_kernel<1','psd_ts_kernel<2'):
    rows = [r[0]/1e3 for r in db.execute("select end-start from kernels where name like '%%%s%%'" % nm)]
    small = sorted(r for r in rows if r < 200)
    print('few$few', nm, 'calls', len(rows), 'small', len(small), 'sum small %.1f us' % sum(small), 'median %.1f' % small[len(small)//2], 'min %.1f' % small[0], 'max %.1f' % small[-1])
PY
done
python -m pytest tests/test_hip_cones.py tests/test_hip_fullsize.py -m gpu -q -x 2>&1 | tail -2
