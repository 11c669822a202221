export TMPDIR=/tmp
python -m pytest tests/test_hip_cones.py -m gpu -q -x -k "psd or Psd or possemidef" 2>&1 | tail -2
for t in 1 0; do
HYP_TS_TEAMS=$t python bench.py --config 4 --steps 6 --warmup 2 --cpu-iters 0 > gpurun_out/ex4_$t.json 2>gpurun_out/ex4_$t.err; python -c "
import json; d=json.loads(open('gpurun_out/ex4_$t.json').read()); print('teams $t', d['ms_per_step'], d.get('phases_ms_per_step'))"
done
