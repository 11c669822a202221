"""Generate the FULL-SIZE oracle trajectories of tests/golden/ (run from the repo root, one or more case names, or no argument for
all of them; about 25 minutes of host BLAS on 8 cores and up to 30 GB of memory for config 4):

    python tests/golden/make_golden_fullsize.py [case ...]

  trajectory_fullsize.json    per case: the oracle's iterate rows (columns of trajectory_harness.COLS), the rows of the same
                              oracle on 1-ulp-perturbed copies of the model, the third-order gate margins, and for polymin
                              the interpolation-point choice (the model must be rebuilt identically on the GPU box)
  schur_probe_fullsize.npz    per case and probed iteration: what trajectory_harness.schur_probe keeps of the oracle's
                              n x n Schur matrix (a seeded 64 x 64 sub-block, the diagonal, S V on four probe vectors, norm)

These are the BASELINE.json configurations AS BENCHMARKED (bench.py's generators: trajectory_harness.fullsize_instance), so that
the kernels the headline number times -- psd_ts3 on two column chunks, the split-K syrk, the look-ahead Cholesky, the
super-block solve plan, the side-200 candidate screen -- sit inside an oracle-compared iteration (tests/test_hip_fullsize_trajectory.py).
The reference is pure Julia and cannot run here: the rows are the restatement's output (oracle/solvers.py, pinned by the
reference's identities and known answers at the reference's own tolerance, tests/test_oracle_*.py)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
HERE = os.path.dirname(os.path.abspath(__file__))

# case -> (solver options of the main run, number of perturbed companions, their options, probed iterations)
CASES = {
    "psdfull_5000_200x1_1": ({}, 3, {}, (1, 3)),                           # config 2 as benchmarked: the whole solve, and so are its three companions
    "psdfull_1300_113x1_1": ({}, 3, {}, (1,)),                             # psd_ts3 + solve plan, one column chunk
    "psdfull_2500_160x1_1": ({}, 3, {}, (1,)),
    "cfg4_5000_80x64_1": ({"iter_limit": 5}, 3, {"iter_limit": 5}, (1, 2)),   # config 4 as benchmarked, first five iterations
    "cfg5p_1": ({"iter_limit": 2}, 1, {"iter_limit": 2}, ()),              # config 5, U = 4845, primal form (n = 1)
    "cfg5d_1": ({"iter_limit": 2}, 1, {"iter_limit": 2}, (1,)),            # dual form: 4844 x 4844 Schur matrix
    "cfg5pw_1": ({}, 3, {"iter_limit": 10}, ()),                           # config 5 primal, the WHOLE solve (46 iterations: the last dozen meet Hessians whose Cholesky fails); companions: first 10 iterates
    "cfg5dw_1": ({}, 3, {"iter_limit": 10}, ()),                           # config 5 dual, the WHOLE solve
}


def main(names):
    os.environ["HYP_GOLDEN_REGEN"] = "1"
    import trajectory_harness as T
    from oracle import polyutils
    jf = os.path.join(HERE, "trajectory_fullsize.json")
    nf = os.path.join(HERE, "schur_probe_fullsize.npz")
    out = json.load(open(jf)) if os.path.exists(jf) else {"columns": list(T.COLS), "cases": {}}
    npz = dict(np.load(nf)) if os.path.exists(nf) else {}
    for name in names:
        opts, npert, popts, probe_iters = CASES[name]
        t0 = time.perf_counter()
        polyutils.LAST_KEEP = None
        inst = T.instance(name)
        keep = polyutils.LAST_KEEP
        o = T.oracle_trajectory(inst, probe_iters=probe_iters, **opts)
        print(name, "oracle: %s after %d iterations, %.0f s" % (o["status"], o["iters"], time.perf_counter() - t0), flush=True)
        ps = []
        for j in range(npert):
            pinst = T.perturbed(inst, seed=99 + j)
            ps.append(T.oracle_trajectory(pinst, **popts))
            del pinst
        rec = {"opts": opts, "perturbed_opts": popts, "status": o["status"], "num_iters": o["iters"], "primal_obj": o["p_obj"],
               "rows": o["rows"].tolist(), "perturbed_rows": [p["rows"].tolist() for p in ps],
               "gate_decades": [None if not np.isfinite(g) else round(float(g), 3) for g in o["gate"]],
               "probe_iters": list(probe_iters), "n": int(len(inst[0])), "p": int(len(inst[2])), "q": int(len(inst[4]))}
        if name.startswith("cfg5") and keep is not None and not name.startswith(("cfg5pw", "cfg5dw")):
            rec["interp_keep"] = [int(v) for v in keep]
        out["cases"][name] = rec
        for it, pr in o["probes"].items():
            for f, v in pr.items():
                npz["%s/%d/%s" % (name, it, f)] = v
        del inst, o, ps
        with open(jf, "w") as f:
            json.dump(out, f, indent=0)
        np.savez(nf, **npz)
        print(name, "done in %.0f s" % (time.perf_counter() - t0), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or list(CASES))
