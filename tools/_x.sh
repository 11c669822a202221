export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_hip_cones.py -m gpu -q -x -k "psd or possemidef or beyond or oracle_vs_hip" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_hip_fullsize_configs.py -m gpu -q -x -k "config4 or 4" 2>&1 | tail -2
for t in 1 0 1; do
HYP_TS_TEAMS=$t timeout 300 python bench.py --config 4 --steps 8 --warmup 2 --cpu-iters 0 2>/dev/null | tail -1 > gpurun_out/ex4.json; python -c "
import json; d=json.loads(open('gpurun_out/ex4.json').read()); print('teams $t', d['ms_per_step'], d['phases_ms_per_step']['sqrt_hess_prod'])"
done
