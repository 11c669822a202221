"""For the PSD trajectory cases of tests/test_hip_solver.py: worst ratio of the HIP path's deviation from the oracle to 100x the
oracle's own 1-ulp sensitivity (three perturbed draws), per column -- run under different HYP_POTRF_* switches to tell a change of
rounding from a loss of accuracy:  python tools/diag_traj_bar.py"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import hypatia_jl_amd as H
from oracle import instances as I
from oracle.build import make_model as omodel
from oracle.solvers import Solver as OSolver
import trajectory_harness as T
cases = [(30, [6, 4], 1), (60, [10, 8, 3], 2), (150, [24, 17], 3), (50, [7] * 5, 4), (60, [5, 8, 8, 8, 8, 3], 5), (60, [5, 8, 8, 8, 8, 3], 6), (60, [5, 8, 8, 8, 8, 3], 7),
         (90, [12, 9, 9, 4], 8)]
for n, sides, seed in cases:
    inst = I.psd_blocks(n, sides, seed=seed)
    hs, ht = T.run_trajectory(H.Solver, H.make_model(inst))
    os_, ot = T.run_trajectory(OSolver, omodel(inst))
    pts = [T.run_trajectory(OSolver, omodel(T.perturbed(inst, seed=99 + j)))[1] for j in range(3)]
    k = min(len(ht), len(ot), min(len(p) for p in pts))
    kp = T.stable_prefix(ot[:k], [p[:k] for p in pts])
    worst1, worst3 = 0.0, 0.0
    for col in (0, 1, 7, 5, 3, 4):
        scale = np.abs(ot[:kp, col]) + (1e-300 if col in (0, 1, 5, 7) else 1e-6)
        dev = np.abs(ht[:kp, col] - ot[:kp, col]) / scale
        f1 = np.abs(pts[0][:kp, col] - ot[:kp, col]) / scale
        f3 = np.max([np.abs(p[:kp, col] - ot[:kp, col]) / scale for p in pts], axis=0)
        worst1 = max(worst1, float(np.max(dev / (100 * np.maximum.accumulate(np.maximum(f1, 1e-13))))))
        worst3 = max(worst3, float(np.max(dev / (100 * np.maximum.accumulate(np.maximum(f3, 1e-13))))))
    print("n=%d sides=%s seed=%d: prefix %d of %d, same alphas %s, worst dev / bar: one draw %.3f, three draws %.3f" %
          (n, sides, seed, kp, k, bool(np.all(ht[:kp, 8] == ot[:kp, 8])), worst1, worst3), flush=True)
