# scratch batch (rewritten per call)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_hip_switches.py -q -x -k "resident or paired or constant_column" 2>&1 | tail -3
for v in 1 2 3; do python bench.py --steps 100 --cpu-iters 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg2', round(d['ms_per_step'],3), 'ms frac', round(d['roofline']['frac'],3), d['phases_ms_per_step'], d.get('ms_per_kkt_solve'))"; done
python bench.py --config 4 --cpu-iters 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg4', round(d['ms_per_step'],3), 'ms frac', round(d['roofline']['frac'],3), d['phases_ms_per_step'], d.get('ms_per_kkt_solve'), d.get('kkt_solves_per_step'))"
