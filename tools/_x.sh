# scratch batch (rewritten per call)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -n 4 -k "wsos or polymin or cfg5 or config5 or generic_oracle" > gpurun_out/x_pytest.log 2>&1; tail -4 gpurun_out/x_pytest.log
for c in 5p 5d; do
for s in "" _off _b _offb; do
  if [ "$s" = _off -o "$s" = _offb ]; then export HYP_WSOS_TRI=0; else unset HYP_WSOS_TRI; fi
  python bench.py --config $c --cpu-iters 0 2>/dev/null | tail -1 > gpurun_out/x_$c$s.json
done; done
unset HYP_WSOS_TRI
python -c "
import json
for c in ('5p','5d'):
  for s in ('','_off','_b','_offb'):
    d=json.load(open('gpurun_out/x_%s%s.json'%(c,s))); print(c+s, round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['phases_ms_per_step'].items()})"
