export TMPDIR=/tmp
python - > /dev/null 2>&1 <<'P'
import cProfile, pstats, sys, io
sys.argv = ["bench.py", "--steps", "60", "--cpu-iters", "0", "--warmup", "3"]
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(70)
open("gpurun_out/pyprof.txt", "w").write(s.getvalue()[:16000])
P
cut -c1-170 gpurun_out/pyprof.txt | head -90
