export TMPDIR=/tmp
for v in 1 0; do
HYP_POTRF_HOSTTIME=1 HYP_PLAN_OVERLAP=$v timeout 300 python bench.py --steps 6 --warmup 2 --cpu-iters 0 > /dev/null 2> gpurun_out/err_$v.txt; grep -c "" gpurun_out/err_$v.txt; grep "potrf host" gpurun_out/err_$v.txt | tail -3
done
