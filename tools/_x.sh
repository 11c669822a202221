# scratch batch (rewritten per call)
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -n 4 -k "dense or gemm or epinorm or matrixcompletion or mc_ or trajectory or fullsize_configs" > gpurun_out/x_pytest.log 2>&1; tail -3 gpurun_out/x_pytest.log
for s in "" _off; do
  if [ "$s" = _off ]; then export HYP_GEMM_AUTOSPLIT_MAX=128; else unset HYP_GEMM_AUTOSPLIT_MAX; fi
  python bench.py --config 3b --cpu-iters 0 2>/dev/null | tail -1 > gpurun_out/x_3b$s.json
done
unset HYP_GEMM_AUTOSPLIT_MAX
python bench.py --config 5p --cpu-iters 0 2>/dev/null | tail -1 > gpurun_out/x_5p.json
python -c "
import json
for c,s in (('3b',''),('3b','_off'),('5p','')):
    d=json.load(open('gpurun_out/x_%s%s.json'%(c,s))); print(c+s, round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['phases_ms_per_step'].items()})"
