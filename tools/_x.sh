export TMPDIR=/tmp
python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -25
for c in 3b 5d; do echo "cfg=$c"; python bench.py --config $c 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['steps'], d['config']['final_status'], d['phases_ms_per_step'])"; done
