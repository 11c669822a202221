export TMPDIR=/tmp
for sd in 100 200 300; do S=$(date +%s); python bench.py --config 3c --mc-side $sd 2> gpurun_out/3c_$sd.err | tail -1 > gpurun_out/3c_$sd.json; E=$(date +%s); echo "wall $((E-S)) s"; python -c "
import json; d=json.loads(open('gpurun_out/3c_$sd.json').read()); print($sd, d['ms_per_step'], d['steps'], d['config']['final_status'], d['config']['n'], d['phases_ms_per_step'], d['setup_s'], d['roofline']['executed_frac'])"; tail -3 gpurun_out/3c_$sd.err; done
