export TMPDIR=/tmp
# scratch script for one-off gpurun calls (tools/_run_gpu.sh = the measurement batch, tools/_run_gpu_tests.sh = the GPU suite)
timeout 600 python bench.py --steps 40 --warmup 10 --cpu-iters 0 | tail -1
