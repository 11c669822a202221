"""Diagnosis: per-iterate relative deviation of the HIP trajectory from the oracle's (computed live on this machine) next to the
deviation of the oracle on the 1-ulp-perturbed model.   python tools/diag_traj_dev.py NAME [iter_limit]"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import trajectory_harness as T
name = sys.argv[1]
opts = {"iter_limit": int(sys.argv[2])} if len(sys.argv) > 2 else {}
inst = T.instance(name)
o = T.oracle_trajectory(inst, **opts)
p = T.oracle_trajectory(T.perturbed(inst), **opts)
h = T.hip_trajectory(name, {k: os.environ[k] for k in T.REFERENCE_ROUTE if k in os.environ} , **opts)
O, P, H = o["rows"], p["rows"], h["rows"]
k = min(len(O), len(P), len(H))
print(name, "iters oracle/pert/hip", len(O) - 1, len(P) - 1, len(H) - 1)
print("it  mu        alpha(o/p/h)      dev_hip: p_obj   mu      tau   |  dev_pert: p_obj   mu     tau")
for i in range(k):
    d = lambda A, c: abs(A[i, c] - O[i, c]) / (abs(O[i, c]) + 1e-300)
    print("%2d %.2e  %.3g/%.3g/%.3g   %.1e %.1e %.1e | %.1e %.1e %.1e" % (i, O[i, 7], O[i, 8], P[i, 8], H[i, 8], d(H, 0), d(H, 7), d(H, 5), d(P, 0), d(P, 7), d(P, 5)))
