python -m pytest tests/test_hip_distributed.py tests/test_c_abi.py -q -x -m gpu 2>&1 | tail -5
(time python bench.py --config 4 > gpurun_out/r02_bench_cfg4_1gpu.json 2> gpurun_out/r02_bench_cfg4_1gpu.err) 2>&1 | grep real
tail -2 gpurun_out/r02_bench_cfg4_1gpu.err; cat gpurun_out/r02_bench_cfg4_1gpu.json
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
(time HYP_FORCE_DIST=1 python bench.py --config 4 --steps 10 > gpurun_out/r02_bench_cfg4_rccl1.json 2> gpurun_out/r02_bench_cfg4_rccl1.err) 2>&1 | grep real
tail -3 gpurun_out/r02_bench_cfg4_rccl1.err; cat gpurun_out/r02_bench_cfg4_rccl1.json
unset MASTER_ADDR MASTER_PORT RANK WORLD_SIZE LOCAL_RANK
(time python bench.py > gpurun_out/r02_bench1.json 2> gpurun_out/r02_bench1.err) 2>&1 | grep real
tail -2 gpurun_out/r02_bench1.err; cat gpurun_out/r02_bench1.json
