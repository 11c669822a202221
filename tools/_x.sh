# scratch batch (rewritten per call): the whole GPU suite and one line per configuration on the round's last commit
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu_head.log 2>&1; tail -2 gpurun_out/pytest_gpu_head.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
rm -f gpurun_out/head_configs.jsonl
python bench.py 2>/dev/null | tail -1 >> gpurun_out/head_configs.jsonl
for c in 3b 5p 5d; do python bench.py --config $c --cpu-iters 0 2>/dev/null | tail -1 >> gpurun_out/head_configs.jsonl; done
python -c "
import json
for l in open('gpurun_out/head_configs.jsonl'): d=json.loads(l); print(d['config']['workload'][:60], round(d['ms_per_step'],3), d['roofline'].get('frac'))"
