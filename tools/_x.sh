# scratch: ts5 RAW A/B
export TMPDIR=/tmp
for f in 1 0; do HYP_TS5_RAW=$f timeout 900 python -m pytest tests/test_hip_cones.py -q -x -m gpu -k "one_wavefront" 2>&1 | tail -2; done
for f in 1 0 1 0; do HYP_TS5_RAW=$f timeout 900 python bench.py --config 4 --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg4 HYP_TS5_RAW=$f', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['phases_ms_per_step'].items()})"; done 2>&1 | tee gpurun_out/bench_ab_ts5raw.txt
