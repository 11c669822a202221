# scratch batch (rewritten per call)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -n 4 -k "epinorm or matrixcompletion or wsos or polymin or mc_ or ens or spectral or dual_feas" > gpurun_out/x_pytest.log 2>&1; tail -5 gpurun_out/x_pytest.log
for c in 3b 5p 5d; do
  python bench.py --config $c --cpu-iters 0 2>/dev/null | tail -1 > gpurun_out/x_$c.json
  HYP_ENS_DUAL_DECIDE=0 HYP_WSOS_PAR=0 python bench.py --config $c --cpu-iters 0 2>/dev/null | tail -1 > gpurun_out/x_${c}_off.json
  python bench.py --config $c --cpu-iters 0 2>/dev/null | tail -1 > gpurun_out/x_${c}_b.json
done
python -c "
import json
for c in ('3b','5p','5d'):
  for s in ('','_off','_b'):
    d=json.load(open('gpurun_out/x_%s%s.json'%(c,s))); print(c+s, round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['phases_ms_per_step'].items()})"
HYP_JACOBI_DBG=1 python bench.py --config 3b --cpu-iters 0 2> gpurun_out/x_3b_jacobi.err >/dev/null
