"""Where does the HOST spend its time in a config-2 iteration?  cProfile over timed iterations of bench.py's headline loop:
python tools/prof_host.py [steps] -> top functions by own time and by cumulative time, ms per iteration."""
import cProfile, io, pstats, sys, types
sys.path.insert(0, ".")
import numpy as np
import bench
import hypatia_jl_amd as H
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
inst = bench.gen_instance(5000, [200], 1)
solver = H.Solver(verbose=False)
solver.load(H.make_model(inst))
solver.setup()
def step():
    while not solver.iterate():
        solver.reset_iterate()
from hypatia_jl_amd.solvers import _blas_limit
cap = _blas_limit(); cap.__enter__()
for _ in range(20):
    step()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    ps = pstats.Stats(pr, stream=s).sort_stats(key)
    ps.print_stats(28)
    txt = s.getvalue()
    print("==== by %s (totals over %d iterations; divide by %d for seconds per iteration)" % (key, steps, steps))
    print("\n".join(l for l in txt.splitlines() if l.strip())[:6000])
