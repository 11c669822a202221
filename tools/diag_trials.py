"""Diagnosis: line-search trials of the HIP path (host-composed, HYP_NO_NATIVE=1, so that every cone test is a call of its own) next
to the oracle's, trial by trial: feasibility verdicts, the check_numerics scalars and the proximity value of every cone.
    HYP_NO_NATIVE=1 python tools/diag_trials.py NAME [first_iter]
prints the first trial at which the two walks differ (a test helper: imports oracle/)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HYP_NO_NATIVE", "1")

import trajectory_harness as T   # noqa: E402

EPS = np.finfo(float).eps


def make_logged(log, solver_ref):
    def check(model, stepper):
        searcher = stepper.searcher
        cand = stepper.temp
        cones = model.cones
        rec = dict(it=solver_ref[0].num_iters, tau=cand.tau, kap=cand.kap, cones=[])
        log.append(rec)
        szk = searcher.szk
        pb = searcher.prox_bound ** 2
        taukap = cand.tau * cand.kap
        if min(cand.tau, cand.kap, taukap) < EPS:
            rec["why"] = "taukap"; return False
        for k in range(len(cones)):
            szk[k] = cand.primal_views[k] @ cand.dual_views[k]
            if szk[k] < EPS:
                rec["why"] = "szk"; return False
        mu = (np.sum(szk) + taukap) / searcher.nup1
        rec["mu"] = mu
        if mu < EPS:
            rec["why"] = "mu"; return False
        tr = taukap / mu
        if tr < searcher.min_prox:
            rec["why"] = "taukap_rel"; return False
        tp = (tr - 1) ** 2
        if tp > pb:
            rec["why"] = "taukap_prox"; return False
        for k in range(len(cones)):
            nu_k = cones[k].get_nu()
            r = szk[k] / (mu * nu_k)
            if r < searcher.min_prox or nu_k * (r - 1) ** 2 > pb:
                rec["why"] = "sz_rel %d" % k; return False
        irtmu = 1.0 / np.sqrt(mu)
        agg = tp
        for k in range(len(cones)):
            c = cones[k]
            c.load_point(cand.primal_views[k], irtmu)
            c.load_dual_point(cand.dual_views[k])
            c.reset_data()
            cr = dict(k=k)
            rec["cones"].append(cr)
            cr["feas"] = bool(c.is_feas())
            if not cr["feas"]:
                rec["why"] = "feas %d" % k; return False
            cr["dual_feas"] = bool(c.is_dual_feas())
            if not cr["dual_feas"]:
                rec["why"] = "dual_feas %d" % k; return False
            cr["numerics"] = bool(c.check_numerics())
            if not cr["numerics"]:
                rec["why"] = "numerics %d" % k; return False
            p = c.get_proxsqr(irtmu, searcher.use_max_prox)
            cr["prox"] = float(p)
            fb = getattr(c, "hess_fact_fallback", None)
            if fb is not None:
                cr["fallback"] = fb() if callable(fb) else fb
            agg = max(agg, p) if searcher.use_max_prox else agg + p
            if not agg < pb:
                rec["why"] = "prox %d" % k; return False
        searcher.prox = np.sqrt(agg)
        rec["why"] = "accepted"
        return True
    return check


def run(mod, solver_cls, model, opts):
    log = []
    ref = [None]
    s = solver_cls(**opts)
    ref[0] = s
    mod.check_cone_points = make_logged(log, ref)
    s.load(model)
    s.solve()
    return s, log


if __name__ == "__main__":
    name = sys.argv[1]
    import hypatia_jl_amd as H
    from hypatia_jl_amd import solvers as hsolv
    from oracle import solvers as osolv
    from oracle.build import make_model as omodel
    inst = T.instance(name)
    hs, hl = run(hsolv, H.Solver, H.make_model(inst), {})
    os_, ol = run(osolv, osolv.Solver, omodel(inst), {})
    print("HIP", hs.status, hs.num_iters, "oracle", os_.status, os_.num_iters)
    for i, (a, b) in enumerate(zip(hl, ol)):
        da = [(c.get("feas"), c.get("dual_feas"), c.get("numerics")) for c in a["cones"]]
        db = [(c.get("feas"), c.get("dual_feas"), c.get("numerics")) for c in b["cones"]]
        if a["why"] != b["why"] or da != db or a["it"] != b["it"]:
            print("first differing trial: #%d" % i)
            for j in range(max(0, i - 3), min(len(hl), len(ol), i + 2)):
                print("  trial", j, "\n    HIP   ", hl[j], "\n    oracle", ol[j])
            break
    else:
        print("all %d trials agree in their verdicts" % min(len(hl), len(ol)))
        worst = 0.0
        for a, b in zip(hl, ol):
            for ca, cb in zip(a["cones"], b["cones"]):
                if "prox" in ca and "prox" in cb:
                    worst = max(worst, abs(ca["prox"] - cb["prox"]) / (abs(cb["prox"]) + 1e-300))
        print("worst relative proximity deviation", worst)
