# scratch: one-launch triangular sweeps, second pass
export TMPDIR=/tmp
for f in 0 1 2; do echo "HYP_TRSV_ONE_LAUNCH=$f"; HYP_TRSV_ONE_LAUNCH=$f timeout 300 python tools/bench_trsv.py 5000 4845 2250 999 --dump gpurun_out/trsv_$f.npz 2>&1 | tail -5; done > gpurun_out/trsv_ab.txt 2>&1
python - <<'P' >> gpurun_out/trsv_ab.txt 2>&1
import numpy as np
a,b,c=[np.load('gpurun_out/trsv_%d.npz'%i) for i in range(3)]
for k in a.files: print(k, 'chain==per-sweep', np.array_equal(a[k],b[k]), 'chain==fused', np.array_equal(a[k],c[k]), 'maxdiff', np.abs(a[k]-c[k]).max(), 'finite', np.isfinite(c[k]).all())
P
cat gpurun_out/trsv_ab.txt
timeout 1500 python -m pytest tests/test_hip_dense.py tests/test_hip_bunchkaufman.py tests/test_hip_switches.py tests/test_hip_solver.py -q -x -m gpu 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_hip_trajectory.py -q -x -m gpu 2>&1 | tail -3
for c in 5p 5d 3b; do for f in 0 2; do HYP_TRSV_ONE_LAUNCH=$f timeout 600 python bench.py --config $c 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c HYP_TRSV_ONE_LAUNCH=$f', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['phases_ms_per_step'].items()})"; done; done 2>&1 | tee gpurun_out/bench_ab_trsv_other.txt
