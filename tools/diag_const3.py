"""diagnosis: the constant column as a third column of the first paired solve (HYP_CONST_COL3=1) against the default -- per
iteration mu, alpha and the constant solution; python tools/diag_const3.py  (runs both modes in child processes)"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child():
    import hypatia_jl_amd as H
    from oracle import instances as I
    inst = I.psd_blocks(40, [12, 7], seed=5)
    rows = []
    s = H.Solver(verbose=False)
    def cb(sv):
        sc = np.array(sv.syssolver.sol_const.vec) if hasattr(sv.syssolver, "sol_const") else np.zeros(1)
        rows.append(dict(mu=float(sv.mu), alpha=float(getattr(sv.stepper, "prev_alpha", 1.0)), sc_norm=float(np.linalg.norm(sc)), sc=sc.tolist(),
                         x_feas=float(sv.x_feas), z_feas=float(sv.z_feas), wres=float(getattr(sv, "worst_dir_res", 0.0)), nsol=int(getattr(sv, "n_solves", 0)),
                         cutoff=float(getattr(sv, "res_norm_cutoff", 0.0))))
    s.iter_callback = cb
    s.load(H.make_model(inst))
    s.solve()
    print(json.dumps(dict(status=s.status, iters=s.num_iters, rows=rows)))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    child()
    sys.exit(0)
res = {}
for mode in ("0", "1"):
    env = dict(os.environ); env["HYP_CONST_COL3"] = mode
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
a, b = res["0"], res["1"]
print("iterations", a["iters"], b["iters"], a["status"], b["status"])
for i in range(min(len(a["rows"]), len(b["rows"]))):
    ra, rb = a["rows"][i], b["rows"][i]
    sa, sb_ = np.array(ra["sc"]), np.array(rb["sc"])
    d = np.linalg.norm(sa - sb_) / max(np.linalg.norm(sa), 1e-300) if sa.shape == sb_.shape else float("nan")
    print("it %2d  mu %.3e / %.3e  alpha %.4f / %.4f  |sc| %.6e / %.6e  rel diff of sol_const %.2e  xfeas %.2e/%.2e  worst_dir_res %.2e/%.2e cutoff %.1e n_solves %d/%d" % (
        i, ra["mu"], rb["mu"], ra["alpha"], rb["alpha"], ra["sc_norm"], rb["sc_norm"], d, ra["x_feas"], rb["x_feas"], ra["wres"], rb["wres"], ra["cutoff"], ra["nsol"], rb["nsol"]))
