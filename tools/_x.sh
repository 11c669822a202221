export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_hip_switches.py -q -x -k "resident or paired or accept_time" 2>&1 | tail -2
for v in 1 0 1 0; do HYP_LATE_UPLOAD=$v python bench.py --steps 300 --cpu-iters 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('late upload=$v cfg2', round(d['ms_per_step'],3))"; done
for v in 1 0 1 0; do HYP_LATE_UPLOAD=$v python bench.py --config 4 --steps 20 --cpu-iters 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('late upload=$v cfg4', round(d['ms_per_step'],3))"; done
