#!/bin/bash
# scratch for one-off gpurun calls (`gpurun --timeout N -- 'bash tools/_x.sh'`); the round's standard batch is tools/_run_gpu.sh
cd /root/repo; export TMPDIR=/tmp
python bench.py > gpurun_out/bench_cfg2_1gpu.json 2> gpurun_out/bench_cfg2_1gpu.err; python -c "
import json; d=json.loads(open('gpurun_out/bench_cfg2_1gpu.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_sqrt_hess_prod'], d['roofline_cholesky'], d['cpu_baseline']['value'])"
python -m pytest tests/test_hip_distributed.py -m gpu -q -x -k "bench" 2>&1 | tail -2
