export TMPDIR=/tmp
rm -f gpurun_out/other_configs.jsonl
for c in 3b 5p 5d; do python bench.py --config $c 2>/dev/null | tail -1 >> gpurun_out/other_configs.jsonl; done
python -c "
import json
for l in open('gpurun_out/other_configs.jsonl'): d=json.loads(l); r=d['roofline']; print(d['config']['workload'][:50], round(d['ms_per_step'],2), 'frac', round(r['frac'],4), 'exec', round(r['executed_frac'],4), r['per_step'])"
