export TMPDIR=/tmp
S=$(date +%s); timeout 1500 python bench.py --config 3c --mc-side 350 2> gpurun_out/3c_350.err | tail -1 > gpurun_out/3c_350.json; E=$(date +%s); echo "wall $((E-S)) s"
python -c "
import json; d=json.loads(open('gpurun_out/3c_350.json').read()); print(350, d['ms_per_step'], d['steps'], d['config']['final_status'], d['config']['n'], d['phases_ms_per_step'], d['setup_s'], d['roofline']['executed_frac'])"; tail -3 gpurun_out/3c_350.err
