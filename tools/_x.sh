# scratch batch (rewritten per call)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -n 4 -k "epinorm or matrixcompletion or mc_ or ens or spectral or trajectory" > gpurun_out/x_pytest.log 2>&1; tail -4 gpurun_out/x_pytest.log
for s in "" _off _b _offb; do
  if [ "$s" = _off -o "$s" = _offb ]; then export HYP_ENS_FUSED_DDER3=0; else unset HYP_ENS_FUSED_DDER3; fi
  python bench.py --config 3b --cpu-iters 0 2>/dev/null | tail -1 > gpurun_out/x_3b$s.json
done
unset HYP_ENS_FUSED_DDER3
python -c "
import json
for s in ('','_off','_b','_offb'):
    d=json.load(open('gpurun_out/x_3b%s.json'%s)); print('3b'+s, round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['phases_ms_per_step'].items()})"
rm -rf /tmp/px_3b; cd /tmp; rocprofv3 --kernel-trace -d /tmp/px_3b -o b -- python $R/bench.py --config 3b --cpu-iters 0 > /dev/null 2>&1; cd $R
DB=$(find /tmp/px_3b -name "*.db" | head -1); python tools/rocpd_stats.py $DB 2>/dev/null | head -40 > gpurun_out/x_3b_stats.csv; grep -E "dder3|grad_aux|feas_fused|hess_prod_fused|closed_inv|jacobi" gpurun_out/x_3b_stats.csv | cut -c1-60,100-170
