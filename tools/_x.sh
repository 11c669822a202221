# scratch batch (rewritten per call)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -n 4 -k "dense or gemm or wsos or polymin or potrf or epinorm" > gpurun_out/x_pytest.log 2>&1; tail -4 gpurun_out/x_pytest.log
for c in 5p 5d 3b; do
for s in "" _off _b _offb; do
  if [ "$s" = _off -o "$s" = _offb ]; then export HYP_GEMM_XCD=0; else unset HYP_GEMM_XCD; fi
  python bench.py --config $c --cpu-iters 0 2>/dev/null | tail -1 > gpurun_out/x_$c$s.json
done; done
unset HYP_GEMM_XCD
python bench.py --cpu-iters 0 2>/dev/null | tail -1 > gpurun_out/x_2.json
HYP_GEMM_XCD=0 python bench.py --cpu-iters 0 2>/dev/null | tail -1 > gpurun_out/x_2_off.json
python -c "
import json
for c in ('5p','5d','3b'):
  for s in ('','_off','_b','_offb'):
    d=json.load(open('gpurun_out/x_%s%s.json'%(c,s))); print(c+s, round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['phases_ms_per_step'].items()})
for s in ('','_off'):
    d=json.load(open('gpurun_out/x_2%s.json'%s)); print('2'+s, round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['phases_ms_per_step'].items()})"
python tools/bench_potrf.py 2>&1 | tail -4
HYP_GEMM_XCD=0 python tools/bench_potrf.py 2>&1 | tail -4
