export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_solver.py tests/test_hip_trajectory.py tests/test_hip_distributed.py tests/test_golden.py tests/test_hip_fullsize.py -m gpu -q -x --tb=short 2>&1 | tail -6
for v in 1 0; do echo "CONST_COL3=$v"; HYP_CONST_COL3=$v python bench.py --steps 100 --cpu-iters 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms_per_step'])"; done
for v in 1 0; do echo "cfg4 CONST_COL3=$v"; HYP_CONST_COL3=$v python bench.py --config 4 --steps 15 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms_per_step'])"; done
