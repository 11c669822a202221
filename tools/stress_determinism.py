"""Run one small solve N times in THIS process and count distinct outcomes (iteration count, objective): a deterministic path
gives one.  usage: python tools/stress_determinism.py [N] [instance: rosenbrock | mixed]"""
import collections
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import hypatia_jl_amd as H
from oracle import instances as I
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
which = sys.argv[2] if len(sys.argv) > 2 else "rosenbrock"
if which == "rosenbrock":
    inst = I.polymin_named("rosenbrock", 5, True, True)
else:
    import trajectory_harness as T
    inst = T.instance("mixed_dual_barriers")
c = collections.Counter()
for i in range(N):
    hs = H.Solver(default_tol_relax=10, iter_limit=250)
    hs.load(H.make_model(inst)); hs.solve()
    c[(hs.get_num_iters(), "%.12e" % hs.get_primal_obj())] += 1
print(which, "distinct outcomes:", len(c), sorted(c.values(), reverse=True))
