# scratch batch (rewritten per call): last sanity run of the round's tree
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python -m pytest tests -m gpu -q -n 4 -k "test_hip_dense or test_hip_cones or test_hip_solver or test_capi" 2>&1 | tail -2
python bench.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench.py default:', d['metric'], round(d['value'],2), d['unit'], round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],3), 'cpu_baseline' in d)"
