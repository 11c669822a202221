export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_switches.py -m gpu -q -x -k "constant_column" 2>&1 | tail -3
timeout 2400 python -m pytest tests/test_hip_dense.py tests/test_hip_solver.py tests/test_hip_trajectory.py tests/test_hip_distributed.py tests/test_hip_fullsize.py tests/test_hip_fullsize_configs.py tests/test_hip_cones.py -m gpu -q -x 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for v in 1 0 1 0; do
HYP_CONST_COL3=$v python bench.py --steps 40 --cpu-iters 0 > gpurun_out/ex.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/ex.json').read()); print('col3 $v', d['ms_per_step'], d['phases_ms_per_step']['update_lhs'], d['phases_ms_per_step']['get_directions'], d['phases_ms_per_step']['search'], d['kkt_solves_per_step'])"
done
