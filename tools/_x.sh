export TMPDIR=/tmp
rm -f gpurun_out/other_configs.jsonl
for c in 3b 5p 5d; do python bench.py --config $c 2>/dev/null | tail -1 >> gpurun_out/other_configs.jsonl; done
python -c "
import json
for l in open('gpurun_out/other_configs.jsonl'): d=json.loads(l); print(d['config']['workload'][:50], d['ms_per_step'], d['roofline']['frac'], d['roofline']['executed_frac'], d['roofline']['per_step'])"
timeout 1500 python -m pytest tests/test_hip_switches.py -m gpu -q -x 2>&1 | tail -2
python bench.py --steps 60 --cpu-iters 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms_per_step'])"
