# scratch batch (rewritten per call)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -n 4 -x -k "dense or gemm or potrf or wsos or polymin or generic_oracle or switches" > gpurun_out/x_pytest.log 2>&1; tail -4 gpurun_out/x_pytest.log
for c in 5p 5d 3b; do
for s in "" _off _b _offb; do
  if [ "$s" = _off -o "$s" = _offb ]; then export HYP_GEMM_KB=16; else unset HYP_GEMM_KB; fi
  python bench.py --config $c --cpu-iters 0 2>/dev/null | tail -1 > gpurun_out/x_$c$s.json
done; done
unset HYP_GEMM_KB
for v in 32 16 32 16; do HYP_GEMM_KB=$v python bench.py --cpu-iters 0 --steps 120 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 kb $v:', round(d['ms_per_step'],3), {k:round(x,2) for k,x in d['phases_ms_per_step'].items()})"; done
python -c "
import json
for c in ('5p','5d','3b'):
  for s in ('','_off','_b','_offb'):
    d=json.load(open('gpurun_out/x_%s%s.json'%(c,s))); print(c+s, round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['phases_ms_per_step'].items()})"
for v in 32 16; do HYP_GEMM_KB=$v python tools/bench_potrf.py 2>&1 | tail -3 | sed "s/^/kb $v: /"; done
