export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_cones.py tests/test_hip_dense.py tests/test_hip_bunchkaufman.py tests/test_hip_trajectory.py tests/test_c_abi.py -m gpu -q -x --tb=short 2>&1 | tail -6
for c in 3b 5d; do python bench.py --config $c 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c', d['ms_per_step'], d['steps'], d['config']['final_status'], d['phases_ms_per_step'])"; done
