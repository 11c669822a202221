export TMPDIR=/tmp
HYP_PROFILE=1 python bench.py --config 4 --steps 20 2> gpurun_out/cfg4_single.err | tail -1 > gpurun_out/bench_cfg4_1gpu.json
HYP_PROFILE=1 HYP_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --config 4 --no-secondary --steps 20 2> gpurun_out/cfg4_rccl.err | tail -1 > gpurun_out/bench_cfg4_rccl_world1.json
grep -v "amdgpu.ids\|socket.cpp\|c10d_logger\|return func" gpurun_out/cfg4_single.err | tail -40
echo =====
grep -v "amdgpu.ids\|socket.cpp\|c10d_logger\|return func" gpurun_out/cfg4_rccl.err | tail -60
