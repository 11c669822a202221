export TMPDIR=/tmp
for v in "1 127" "0 127" "1 159" "1 97" "1 127"; do set -- $v
HYP_TS_PERSIST=$1 HYP_TS_WPX=$2 timeout 300 python bench.py --steps 40 --cpu-iters 0 > gpurun_out/ex.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/ex.json').read()); print('persist $1 wpx $2', d['ms_per_step'], d['phases_ms_per_step']['sqrt_hess_prod'], d['phases_ms_per_step']['syrk'])"
done
