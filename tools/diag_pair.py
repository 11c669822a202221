"""Diagnosis: at iterate K of a named instance, the four directions of the step through the paired device solve against the
single-column device solve, each with its true KKT residual (apply_lhs).   python tools/diag_pair.py NAME K"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import trajectory_harness as T
import hypatia_jl_amd as H
from hypatia_jl_amd import solvers as HS
name, K = sys.argv[1], int(sys.argv[2])
inst = T.instance(name)
hs = H.Solver(iter_limit=K)
hs.load(H.make_model(inst)); hs.solve()
st, sysv = hs.stepper, hs.syssolver
print("mu", hs.mu, "p", hs.model.p)
sysv.update_lhs(hs)
singles, res_s = [], []
for k, upd in enumerate((HS.update_rhs_cent, HS.update_rhs_pred)):
    upd(hs, st.rhs)
    st.rhs2[k] = st.rhs.vec
    hs.worst_dir_res = 0.0
    HS.get_directions(st, hs)
    singles.append(st.dir.vec.copy())
    HS.apply_lhs(st, hs)
    res_s.append(np.max(np.abs(st.temp.vec - st.rhs.vec)))
(ra, rb), ns = sysv.get_directions2_native(hs, st.dir2, st.rhs2)
print("pair reported residuals", ra, rb, "solves", ns)
for k in range(2):
    st.dir.vec[:] = st.dir2[k]
    st.rhs.vec[:] = st.rhs2[k]
    HS.apply_lhs(st, hs)
    rp = np.max(np.abs(st.temp.vec - st.rhs.vec))
    print("dir", k, "single true res %.3e  pair true res %.3e  |pair - single| / |single| = %.3e" % (res_s[k], rp, np.linalg.norm(st.dir2[k] - singles[k]) / np.linalg.norm(singles[k])))

# ---- the oracle at the SAME point: its directions against both device variants
from oracle import solvers as OS
from oracle.build import make_model as omodel
os_ = OS.Solver(iter_limit=K)
os_.load(omodel(inst)); os_.solve()
print("point deviation HIP vs oracle before overwrite: %.3e" % (np.linalg.norm(hs.point.vec - os_.point.vec) / np.linalg.norm(os_.point.vec)))
os_.point.vec[:] = hs.point.vec
os_.calc_mu()
for k in range(len(os_.model.cones)):   # cones at the scaled point, as Solvers.jl does before a step
    pass
ost = os_.stepper
# reload the cones exactly as the stepper does at the start of a step (combined.jl:53-64)
OS_step_prep = getattr(OS, "load_cones_at_point", None)
rtmu = np.sqrt(os_.mu); irtmu = 1.0 / rtmu
for k, cone in enumerate(os_.model.cones):
    cone.load_point(os_.point.primal_views[k], irtmu)
    cone.load_dual_point(os_.point.dual_views[k])
    cone.reset_data()
    assert cone.is_feas()
    cone.get_grad()
os_.syssolver.update_lhs(os_)
for k, (upd, oupd) in enumerate(((HS.update_rhs_cent, OS.update_rhs_cent), (HS.update_rhs_pred, OS.update_rhs_pred))):
    oupd(os_, ost.rhs)
    n0 = os_.n_solves if hasattr(os_, "n_solves") else 0
    OS.get_directions(ost, os_)
    d_orc = ost.dir.vec.copy()
    ns = (os_.n_solves - n0) if hasattr(os_, "n_solves") else -1
    sc = np.linalg.norm(d_orc)
    print("dir", k, "oracle solves", ns, " |single - oracle| %.3e  |pair - oracle| %.3e  (rhs dev %.2e)" % (
        np.linalg.norm(singles[k] - d_orc) / sc, np.linalg.norm(st.dir2[k] - d_orc) / sc, np.linalg.norm(st.rhs2[k] - ost.rhs.vec) / (np.linalg.norm(ost.rhs.vec) + 1e-300)))

# ---- the acceptance test of the third-order terms (steppers/common.jl:41-50, 100-108) on both sides, same point, same direction
def viols(solver, dirvec, mod, which):
    out = []
    st_ = solver.stepper
    st_.dir.vec[:] = dirvec
    irtrtmu = 1.0 / np.sqrt(np.sqrt(solver.mu))
    for k, cone_k in enumerate(solver.model.cones):
        if not cone_k.use_dder3():
            out.append(None); continue
        prim_dir_k = np.array(st_.dir.primal_views[k])
        scal = irtrtmu * prim_dir_k
        Hp = np.zeros_like(scal)
        if which == "pred":
            cone_k.hess_prod_slow(Hp, prim_dir_k)
        else:
            cone_k.hess_prod_slow(Hp, scal)
        d3 = np.array(cone_k.dder3(scal))
        dot1 = d3 @ np.array(cone_k.point)
        dot2 = (irtrtmu if which == "pred" else 1.0) * (scal @ Hp)
        out.append(abs(dot1 - dot2) / (np.sqrt(np.finfo(float).eps) + abs(dot2)))
    return out
for which, dvec in (("cent", singles[0]), ("pred", singles[1])):
    print(which, "dder3_viol HIP   ", ["%.2e" % v if v is not None else None for v in viols(hs, dvec, HS, which)])
    print(which, "dder3_viol oracle", ["%.2e" % v if v is not None else None for v in viols(os_, dvec, OS, which)])
