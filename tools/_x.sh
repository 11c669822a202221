export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_cones.py tests/test_hip_dense.py -m gpu -q -x --tb=short 2>&1 | tail -3
for c in 5p 5d; do python bench.py --config $c 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c', d['ms_per_step'], d['steps'], d['config']['final_status'], d['phases_ms_per_step'], d['roofline']['per_step'])"; done
rocprofv3 --kernel-trace --stats -d /tmp/p5 -o b -- python bench.py --config 5d > /dev/null 2>&1
python tools/rocpd_stats.py $(find /tmp/p5 -name "*.db" | head -1) 2>/dev/null | head -6
