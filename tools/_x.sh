# scratch batch (rewritten per call): a quick sanity run
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 60 --cpu-iters 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), d['unit'], round(d['ms_per_step'],3), 'ms, frac', round(d['roofline']['frac'],3))"
