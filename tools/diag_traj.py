import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hypatia_jl_amd as H
from oracle import instances as I
from oracle.build import make_model as omodel
from oracle.solvers import Solver as OSolver

def traj(cls, model):
    rows = []
    s = cls()
    s.iter_callback = lambda sv: rows.append((sv.primal_obj, sv.mu, sv.x_feas, sv.z_feas, getattr(sv.stepper, "prev_alpha", 1.0), sv.worst_dir_res))
    s.load(model); s.solve()
    return s, np.array(rows)

for (n, sides, seed) in [(60, [10, 8, 3], 2), (150, [24, 17], 3)]:
    inst = I.psd_blocks(n, sides, seed=seed)
    hs, ht = traj(H.Solver, H.make_model(inst))
    os_, ot = traj(OSolver, omodel(inst))
    rng = np.random.default_rng(99)
    G2 = inst[3] * (1.0 + np.finfo(float).eps * rng.choice([-1.0, 1.0], size=inst[3].shape))
    ps_, pt = traj(OSolver, omodel(inst[:3] + (G2,) + inst[4:]))
    print("case", n, sides, "iters hip/oracle/perturbed:", hs.num_iters, os_.num_iters, ps_.num_iters)
    k = min(len(ht), len(ot), len(pt))
    for i in range(k):
        print(f"{i:3d} mu {ot[i,1]:.3e} | alpha h {ht[i,4]:.4f} o {ot[i,4]:.4f} p {pt[i,4]:.4f} | relmu h {abs(ht[i,1]-ot[i,1])/ot[i,1]:.2e} p {abs(pt[i,1]-ot[i,1])/ot[i,1]:.2e} | dirres h {ht[i,5]:.1e} o {ot[i,5]:.1e}")
