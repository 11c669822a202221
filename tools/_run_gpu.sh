export TMPDIR=/tmp
HYP_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --steps 6 --warmup 1 --secondary-steps 10 2>/dev/null | tail -1 | cut -c1-200
HYP_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --config 2 --steps 6 --warmup 1 2>/dev/null | tail -1 | cut -c1-200
python bench.py --steps 10 --cpu-iters 0 2>/dev/null | tail -1 | cut -c1-120
python bench.py --config 3b 2>/dev/null | tail -1 | cut -c1-120
