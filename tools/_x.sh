# scratch batch (rewritten per call)
export TMPDIR=/tmp
for c in 5p 5d; do
for s in "" _off _b _offb; do
  if [ "$s" = _off -o "$s" = _offb ]; then unset HYP_WSOS_TILE; else export HYP_WSOS_TILE=128; fi
  python bench.py --config $c --cpu-iters 0 2>/dev/null | tail -1 > gpurun_out/x_$c$s.json
done; done
unset HYP_WSOS_TILE
python -c "
import json
for c in ('5p','5d'):
  for s in ('','_off','_b','_offb'):
    d=json.load(open('gpurun_out/x_%s%s.json'%(c,s))); print(c+s, '(128 tiles)' if s in ('','_b') else '(default)', round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['phases_ms_per_step'].items()})"
HYP_WSOS_TILE=128 python -m pytest tests -m gpu -q -n 4 -x -k "wsos and (generic_oracle or identities)" 2>&1 | tail -2
