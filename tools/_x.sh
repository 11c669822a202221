# scratch batch (rewritten per call)
export TMPDIR=/tmp
HYP_PROFILE=1 python bench.py --config 3b --cpu-iters 0 > gpurun_out/x_prof_3b.out 2> gpurun_out/x_prof_3b.err
HYP_PROFILE=1 python bench.py --cpu-iters 0 > gpurun_out/x_prof_2.out 2> gpurun_out/x_prof_2.err
tail -3 gpurun_out/x_prof_3b.err | cut -c1-300
