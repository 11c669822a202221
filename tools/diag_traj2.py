"""Print the line-search step sizes of the HIP path, the oracle and a 1-ulp-perturbed oracle side by side."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import hypatia_jl_amd as H
from oracle import instances as I
from oracle.build import make_model as omodel
from oracle.solvers import Solver as OSolver
from test_hip_solver import _trajectory

n, sides, seed = 30, [6, 4], 1
inst = I.psd_blocks(n, sides, seed=seed)
hs, ht = _trajectory(H.Solver, H.make_model(inst))
os_, ot = _trajectory(OSolver, omodel(inst))
pts = []
for s in (99, 7, 8):
    rng = np.random.default_rng(s)
    G2 = inst[3] * (1.0 + np.finfo(float).eps * rng.choice([-1.0, 1.0], size=inst[3].shape))
    pts.append(_trajectory(OSolver, omodel(inst[:3] + (G2,) + inst[4:]))[1])
k = min(len(ht), len(ot), *[len(p) for p in pts])
print("it   mu(oracle)   alpha: hip  oracle  pert99 pert7 pert8    relerr mu hip")
for i in range(k):
    print("%2d  %.3e   %6.4f %6.4f %6.4f %6.4f %6.4f   %.2e" % (i, ot[i, 7], ht[i, 8], ot[i, 8], pts[0][i, 8], pts[1][i, 8], pts[2][i, 8],
                                                              abs(ht[i, 7] - ot[i, 7]) / ot[i, 7]))
