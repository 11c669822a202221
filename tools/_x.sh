# scratch batch (rewritten per call): host affinity near the GPU / on the other NUMA node vs the default
export TMPDIR=/tmp
run() { python bench.py --config $1 --cpu-iters 0 $3 2>gpurun_out/x_pin.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['phases_ms_per_step'].items()})"; grep "host affinity" gpurun_out/x_pin.err | cut -c1-230; }
for i in 1 2; do
  HYP_BENCH_PIN=0 run 5p "5p default (report):"
  HYP_BENCH_PIN=1 run 5p "5p pinned LOCAL    :"
  HYP_BENCH_PIN=far run 5p "5p pinned FAR      :"
done
HYP_BENCH_PIN=1 run 3b "3b pinned LOCAL    :"
HYP_BENCH_PIN=far run 3b "3b pinned FAR      :"
HYP_BENCH_PIN=1 run 2 "cfg2 pinned LOCAL  :" "--steps 120"
HYP_BENCH_PIN=far run 2 "cfg2 pinned FAR    :" "--steps 120"
