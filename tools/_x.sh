export TMPDIR=/tmp
for v in "4 3" "4 4" "3 3"; do set -- $v
rm -rf /tmp/pb; HYP_TS_BT1=$1 HYP_TS_BT2=$2 rocprofv3 --kernel-trace -d /tmp/pb -o b -- python bench.py --steps 12 --warmup 2 --cpu-iters 0 > /dev/null 2>&1
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob('/tmp/pb/**/*.db', recursive=True)[0])
for nm in ('psd_ts3_kernel<1','psd_ts3_kernel<2'):
    rows = [r[0]/1e3 for r in db.execute("select end-start from kernels where name like '%%%s%%'" % nm)]
    big = sorted(r for r in rows if r > 200)
    print('bt $1 $2', nm, 'sum/iter %.1f us' % (sum(big)/14))
PY
done
