"""CPU oracle time per IPM iteration at config 2 for several host BLAS pool sizes."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from threadpoolctl import threadpool_limits
import bench
from oracle.build import make_model as omodel
from oracle.solvers import Solver as OSolver

inst = bench.gen_instance(5000, [200], 1)
for nt in [int(a) for a in sys.argv[1:]] or [1, 16, 64]:
    with threadpool_limits(limits=nt, user_api="blas"):
        s = OSolver(verbose=False, iter_limit=2)
        s.load(omodel(inst))
        t0 = time.perf_counter()
        s.solve()
        print("threads %3d: %.2f s per iteration (total solve %.1f s incl. setup)" % (nt, s.iter_time / max(s.num_iters, 1), time.perf_counter() - t0), flush=True)
