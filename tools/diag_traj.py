"""python tools/diag_traj.py NAME [route-json]: the HIP trajectory of a full-size golden case against the committed oracle rows,
iterate by iterate (relative deviation per column next to the oracle's own 1-ulp sensitivity)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import trajectory_harness as T
name = sys.argv[1]
route = json.loads(sys.argv[2]) if len(sys.argv) > 2 else {}
rec = json.load(open(os.path.join(ROOT, "tests", "golden", "trajectory_fullsize.json")))["cases"][name]
ht = T.hip_trajectory(name, route, **rec["opts"])
O = np.array(rec["rows"]); Ps = [np.array(p) for p in rec["perturbed_rows"]]; H = ht["rows"]
k = min(len(H), len(O))
print(name, route, "HIP", ht["status"], ht["iters"], "oracle", rec["status"], rec["num_iters"], "bk", ht.get("bk_stats"))
print("cols", T.COLS)
for i in range(k):
    kk = min(len(p) for p in Ps)
    devs = []
    for col in (0, 1, 7, 5, 3, 4):
        scale = abs(O[i, col]) + (1e-300 if col in (0, 1, 5, 7) else 1e-6)
        d = abs(H[i, col] - O[i, col]) / scale
        f = max(abs(p[i, col] - O[i, col]) / scale for p in Ps) if i < kk else float("nan")
        devs.append("%.1e/%.1e" % (d, f))
    print("it %2d mu %.2e alpha H %.4g O %.4g | p_obj d_obj mu tau xfeas zfeas (dev/floor): %s" % (i, O[i, 7], H[i, 8], O[i, 8], " ".join(devs)))
