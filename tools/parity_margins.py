"""How far inside the parity bar every full-size fixture sits: for each case of
tests/golden/trajectory_fullsize.json, default route, the HIP trajectory against the committed oracle rows through
tests/trajectory_harness.compare; per column the worst deviation, its ratio to the bar while mu >= 1e-3 (bar = max(1e-10, 3x the
oracle's own 1-ulp sensitivity)), its ratio to the 100x-sensitivity limit on the compared prefix, and the oracle's sensitivity itself.
A ratio creeping towards 1 is a drift to look at before it becomes a failure.   python tools/parity_margins.py [case ...]"""
import json
import os
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import trajectory_harness as T

CASES = json.load(open(os.path.join("tests", "golden", "trajectory_fullsize.json")))["cases"]
names = sys.argv[1:] or sorted(CASES)
print("case                      prefix iters(hip/oracle)  column   worst_dev   dev/bar(mu>=1e-3)  dev/limit(prefix)  oracle_1ulp_sensitivity")
for name in names:
    rec = CASES[name]
    ht = T.hip_trajectory(name, T.DEFAULT_ROUTE, timeout=1700, probe_iters=[], **rec["opts"])
    gate = np.array([np.inf if g is None else g for g in rec["gate_decades"]])
    ot = dict(status=rec["status"], iters=rec["num_iters"], rows=np.array(rec["rows"]), gate=gate)
    pt = dict(rows=[np.array(p) for p in rec["perturbed_rows"]])
    rep = T.compare(ht, ot, pt, label=name)
    for cname, (m_tight, m_loose, fl) in rep["margin"].items():
        print("%-25s %3d    %3d / %3d          %-7s  %.3e   %.3f              %.3f              %.3e" % (
            name, rep["prefix"], rep["iters_hip"], rep["iters_oracle"], cname, rep["worst"][cname], m_tight, m_loose, fl), flush=True)
