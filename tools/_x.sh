export TMPDIR=/tmp
python -m pytest tests/test_hip_cones.py -m gpu -q -x -k "psd or possemidef or beyond" 2>&1 | tail -2
for i in 1 2; do
python bench.py --steps 40 --cpu-iters 0 > gpurun_out/ex.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/ex.json').read()); print(d['ms_per_step'], d['phases_ms_per_step']['sqrt_hess_prod'])"
done
