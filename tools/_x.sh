#!/bin/bash
# scratch for one-off gpurun calls (`gpurun --timeout N -- 'bash tools/_x.sh'`); the round's standard batch is tools/_run_gpu.sh
cd /root/repo; export TMPDIR=/tmp
for v in "" "HYP_POTRF_TINV=0" "HYP_POTRF_MFMA=0"; do echo "== 5p $v"; env $v python bench.py --config 5p 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), d['steps'], d['phases_ms_per_step'], d['roofline']['per_step']['bunch_kaufman_factorizations'], d['roofline']['per_step']['cone_hessian_factorizations'])"; done
