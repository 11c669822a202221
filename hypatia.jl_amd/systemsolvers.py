"""QRCholDenseSystemSolver on the MI355X: host mirror of the reference's SystemSolver interface
(/root/reference/src/Solvers/systemsolvers/qrchol.jl:104-257 and common.jl:129-208).

`load`, `update_lhs`, `solve_subsystem3` are the three methods Hypatia requires of a
`QRCholSystemSolver` subtype; they forward to hyp_sys_* of the C-ABI.  The shared
reductions `solve_system` / `solve_subsystem4` / `setup_rhs3` are kept on the host exactly as the
reference inherits them.
"""
import ctypes
import os
import time

import numpy as np

from . import _lib as L

c_int, c_vp = ctypes.c_int, ctypes.c_void_p


class SubPoint:
    """(x, y, z) point of the 3x3 subsystem (common.jl:184-208)."""

    def __init__(self, model):
        n, p, q = model.n, model.p, model.q
        self.vec = np.zeros(n + p + q)
        self.x = self.vec[:n]
        self.y = self.vec[n:n + p]
        self.z = self.vec[n + p:]
        self.z_views = [self.z[idx] for idx in model.cone_idxs]


def dot_obj(model, point):   # common.jl:210-211
    return model.c @ point.x + model.b @ point.y + model.h @ point.z


class QRCholDenseSystemSolver:
    def __init__(self):
        self._h = None

    def __del__(self):
        try:
            if self._h is not None and L._lib is not None:
                L._lib.hyp_sys_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- qrchol.jl:138-179
    def load(self, solver):
        self.__dict__.pop("_screen_usable", None)   # (the candidate screen is decided per loaded model)
        self._dirs_resident = False
        model = solver.model
        n, p, q = model.n, model.p, model.q
        lib = L.lib()
        if self._h is not None:
            lib.hyp_sys_destroy(self._h)
            self._h = None
        handles = (c_vp * len(model.cones))(*[cone._h for cone in model.cones])
        h = c_vp()
        L.check(lib.hyp_sys_create(L.ctx(), n, p, q, handles, len(model.cones), ctypes.byref(h)), "hyp_sys_create")
        self._h = h
        self.n, self.p, self.q = n, p, q
        G = np.asfortranarray(model.G)
        if p == 0:
            L.check(lib.hyp_sys_load(h, G.ctypes.data_as(c_vp), None, None, None, None), "hyp_sys_load")
        else:
            Q = np.asfortranarray(solver.Ap_Q)    # G * Ap_Q (qrchol.jl:154) is formed on the device
            R = np.asfortranarray(solver.Ap_R)
            L.check(lib.hyp_sys_load(h, G.ctypes.data_as(c_vp), None, None, Q.ctypes.data_as(c_vp), R.ctypes.data_as(c_vp)), "hyp_sys_load")
        self.use_sqrt_hess_cones = [False] * len(model.cones)
        # setup_point_sub (common.jl:184-208)
        self.sol_sub = SubPoint(model)
        self.rhs_sub = SubPoint(model)
        self.rhs_const = SubPoint(model)
        self.sol_const = SubPoint(model)
        self.rhs_const.x[:] = -model.c
        self.rhs_const.y[:] = model.b
        self.rhs_const.z[:] = model.h
        self.last_info = 0
        self.used_fallback = False
        self.fallback_kind = 0
        # device-resident get_directions (hyp_sys_get_directions): model vectors live on the GPU too
        cc, bb, hh = (np.ascontiguousarray(v, dtype=np.float64) for v in (model.c, model.b, model.h))
        AA = np.asfortranarray(model.A, dtype=np.float64) if p > 0 else None
        L.check(lib.hyp_sys_load_model(h, L.vec_ptr(cc), L.vec_ptr(bb), L.vec_ptr(hh),
                                       AA.ctypes.data_as(c_vp) if AA is not None else None), "hyp_sys_load_model")
        self.native_directions = os.environ.get("HYP_NO_NATIVE", "0") in ("", "0")
        return self

    # y = alpha * op(G) x + beta * y on the device-resident model.G
    def mul_G(self, trans, x, alpha=1.0, beta=0.0, y=None):
        xx = np.ascontiguousarray(x, dtype=np.float64)
        ny = self.n if trans else self.q
        if y is None:
            y = np.zeros(ny)
        L.check(L.lib().hyp_sys_mul_G(self._h, int(trans), float(alpha), L.vec_ptr(xx), float(beta), L.vec_ptr(y)), "hyp_sys_mul_G")
        return y

    one_pass_residual_products = True   # calc_convergence_params takes G' z and G x + s from ONE pass over G

    def residual_products(self, pt):
        """G' z, G x + s, h' z and z' s of a point in one call (hyp_sys_residual_products: G is read once for both products)"""
        Gtz, Gx_s, dots = np.zeros(self.n), np.zeros(max(self.q, 1)), np.zeros(2)
        L.check(L.lib().hyp_sys_residual_products(self._h, L.vec_ptr(np.ascontiguousarray(pt.x)), L.vec_ptr(np.ascontiguousarray(pt.z)),
                                                  L.vec_ptr(np.ascontiguousarray(pt.s)), L.vec_ptr(Gtz), L.vec_ptr(Gx_s), L.vec_ptr(dots)),
                "hyp_sys_residual_products")
        return {"Gtz": Gtz, "Gx_s": Gx_s[:self.q], "hz": float(dots[0]), "zs": float(dots[1])}

    # ---- qrchol.jl:181-199
    def update_lhs(self, solver):
        model = solver.model
        if self.native_directions:   # one call: factorization + the constant-column solve, both kept on the device
            nc = len(model.cones)
            flags = (c_int * max(nc, 1))()
            info, fb = c_int(0), c_int(0)
            t0 = time.perf_counter()
            L.check(L.lib().hyp_sys_update_lhs(self._h, flags, ctypes.byref(info), ctypes.byref(fb), L.vec_ptr(self.sol_const.vec)),
                    "hyp_sys_update_lhs")
            solver.time_upfact += time.perf_counter() - t0
            self.use_sqrt_hess_cones = [bool(flags[k]) for k in range(nc)]
            self.last_info, self.used_fallback = info.value, bool(fb.value)
            self.fallback_kind = fb.value   # 0 Cholesky, 1 Bunch-Kaufman, 2 diagonal shift + Bunch-Kaufman
            if info.value != 0:
                print("positive definite linear system factorization failed")
            return self
        self.last_info = 0
        if model.n - model.p > 0:
            self.update_lhs_fact(solver)
            if self.last_info != 0:
                # every link of posdef_fact_copy! failed (qrchol.jl:253-255): there is no factorization to solve the constant
                # column with; the stepper sees last_info and ends in NumericalFailure (combined.jl:97-117)
                return self
        hh = np.ascontiguousarray(model.h)
        L.check(L.lib().hyp_sys_block_hess_prod(self._h, L.vec_ptr(self.rhs_const.z), L.vec_ptr(hh)), "hyp_sys_block_hess_prod")
        self.solve_subsystem3(solver, self.sol_const, self.rhs_const)
        return self

    # ---- the direction phase of CombinedStepper.step in one call (p = 0): update_lhs + 4 right-hand sides + 2 paired solves
    def step_directions_native(self, solver, stepper):
        model = solver.model
        nc = len(model.cones)
        flags = (c_int * max(nc, 1))()
        info, fb, ns = c_int(0), c_int(0), c_int(0)
        resn = (ctypes.c_double * 4)()
        resid = np.concatenate([solver.x_residual, solver.y_residual, solver.z_residual])
        dirs4 = stepper.dirs4
        if not hasattr(self, "_dir_rows_set"):
            # the line search walks the schedule on the directions this call leaves on the device and hands back the accepted candidate's
            # z / tau / s / kap rows (search_alpha_native); update_stepper_points_x then reads only the x rows of the four directions:
            # only those are downloaded (q = 207 360: 13 MB per iteration otherwise).  HYP_DIRS_X_ONLY=0: the whole vectors.
            from .solvers import _cap
            x_only = (os.environ.get("HYP_DIRS_X_ONLY", "1") != "0" and _cap(self, "search") and getattr(self, "cand_in_temp", True)
                      and not getattr(self, "row_local", False) and self._screen_ok())
            L.check(L.lib().hyp_sys_set_direction_rows(self._h, 1 if x_only else 0), "hyp_sys_set_direction_rows")
            self._dir_rows_set = True
        L.check(L.lib().hyp_sys_step_directions(self._h, L.vec_ptr(solver.point.vec), L.vec_ptr(resid), float(solver.tau_residual), float(solver.mu),
                                                int(solver.max_ref_steps), float(solver.res_norm_cutoff), 0.5, dirs4.ctypes.data_as(c_vp), resn,
                                                ctypes.byref(ns), flags, ctypes.byref(info), ctypes.byref(fb), L.vec_ptr(self.sol_const.vec)),
                "hyp_sys_step_directions")
        self.use_sqrt_hess_cones = [bool(flags[k]) for k in range(nc)]
        self.last_info, self.used_fallback = info.value, bool(fb.value)
        self.fallback_kind = fb.value   # 0 Cholesky, 1 Bunch-Kaufman, 2 diagonal shift + Bunch-Kaufman
        if info.value != 0:
            print("positive definite linear system factorization failed")
            return False
        # (stepper.dir_cent / dir_pred / dir_centadj / dir_predadj are views of the rows of dirs4: nothing to copy)
        solver.n_solves += ns.value
        assert not any(np.isnan(resn[k]) for k in range(4))
        if solver.max_ref_steps > 0:
            solver.worst_dir_res = max(solver.worst_dir_res, *[resn[k] for k in range(4)])
        self._dirs_resident = True   # (until CombinedStepper.step is done with its searches: search_alpha_native)
        return True

    def last_update_lhs_seconds(self):
        out = ctypes.c_double(0.0)
        L.check(L.lib().hyp_sys_last_update_lhs_seconds(self._h, ctypes.byref(out)), "hyp_sys_last_update_lhs_seconds")
        return out.value

    # ---- two independent right-hand sides per pass (hyp_sys_get_directions2)
    def get_directions2_native(self, solver, dirs2, rhss2, min_impr_tol=0.5):
        """dirs2 / rhss2: (2 x len(Point.vec)) C-contiguous arrays; returns (res_norms[2], n_solves)"""
        res = (ctypes.c_double * 2)()
        ns = c_int(0)
        L.check(L.lib().hyp_sys_get_directions2(self._h, dirs2.ctypes.data_as(c_vp), rhss2.ctypes.data_as(c_vp), float(solver.mu),
                                                float(solver.point.tau), int(solver.max_ref_steps), float(solver.res_norm_cutoff),
                                                float(min_impr_tol), res, ctypes.byref(ns)), "hyp_sys_get_directions2")
        return (res[0], res[1]), ns.value

    # ---- search.jl:74-138 for all cones in one call; keeps the host mirrors of the reloaded cones in step
    def check_cone_points_native(self, model, cand, searcher):
        acc, nl = c_int(0), c_int(0)
        prox, irtmu = ctypes.c_double(0.0), ctypes.c_double(0.0)
        L.check(L.lib().hyp_sys_check_cone_points(self._h, L.vec_ptr(cand.ztsk), float(searcher.min_prox), float(searcher.prox_bound),
                                                  int(bool(searcher.use_max_prox)), float(searcher.nup1), ctypes.byref(acc),
                                                  ctypes.byref(prox), ctypes.byref(nl), ctypes.byref(irtmu)), "hyp_sys_check_cone_points")
        for k in range(nl.value):
            model.cones[k]._mirror_loaded(cand.primal_views[k], irtmu.value, cand.dual_views[k])
        if acc.value:
            searcher.prox = prox.value
        return bool(acc.value)

    # ---- search.jl:46-69 for one stepper mode in one call (candidates formed natively)
    def search_alpha_native(self, model, point, stepper, sched):
        """sched: 1-based start index into searcher.alpha_sched (as search_alpha); returns (alpha, next prev_sched)"""
        searcher = stepper.searcher
        sc = np.ascontiguousarray(searcher.alpha_sched, dtype=np.float64)
        cand = stepper.temp
        idx, nt, nl = c_int(-1), c_int(0), c_int(0)
        prox, irtmu = ctypes.c_double(0.0), ctypes.c_double(0.0)
        if getattr(self, "_dirs_resident", False) and self._screen_ok():
            # the point and the directions are the ones step_directions_native left on the device (the stepper has not touched
            # them since): the schedule's candidates are formed and screened there, nothing of length q is uploaded
            L.check(L.lib().hyp_sys_search_alpha_resident(
                self._h, int(stepper.unadj_only), int(stepper.cent_only), L.vec_ptr(sc), len(sc), int(sched - 1), float(searcher.min_prox),
                float(searcher.prox_bound), int(bool(searcher.use_max_prox)), float(searcher.nup1), L.vec_ptr(cand.ztsk), ctypes.byref(idx),
                ctypes.byref(prox), ctypes.byref(nt), ctypes.byref(nl), ctypes.byref(irtmu)), "hyp_sys_search_alpha_resident")
        else:
            L.check(L.lib().hyp_sys_search_alpha(
                self._h, L.vec_ptr(point.ztsk), L.vec_ptr(stepper.dir_cent.ztsk), L.vec_ptr(stepper.dir_pred.ztsk),
                L.vec_ptr(stepper.dir_centadj.ztsk), L.vec_ptr(stepper.dir_predadj.ztsk), int(stepper.unadj_only), int(stepper.cent_only),
                L.vec_ptr(sc), len(sc), int(sched - 1), float(searcher.min_prox), float(searcher.prox_bound),
                int(bool(searcher.use_max_prox)), float(searcher.nup1), L.vec_ptr(cand.ztsk), ctypes.byref(idx), ctypes.byref(prox),
                ctypes.byref(nt), ctypes.byref(nl), ctypes.byref(irtmu)), "hyp_sys_search_alpha")
        searcher.n_trials += nt.value
        for k in range(nl.value):
            model.cones[k]._mirror_loaded(cand.primal_views[k], irtmu.value, cand.dual_views[k])
        if idx.value >= 0:
            searcher.prox = prox.value
            searcher.prev_sched = idx.value + 1
            return float(sc[idx.value])
        searcher.prev_sched = len(sc) + 1
        return 0.0

    def _screen_ok(self):
        """the side-by-side candidate screen applies to the loaded model (one PosSemidefTri cone) and HYP_SEARCH_RESIDENT is not 0"""
        if not hasattr(self, "_screen_usable"):
            self.search_screen_stats()
            if os.environ.get("HYP_SEARCH_RESIDENT", "1") == "0":
                self._screen_usable = False
        return self._screen_usable

    def search_screen_stats(self):
        """(screens run, candidates rejected by them) of the side-by-side candidate screen inside search_alpha_native"""
        u, a, b = c_int(0), ctypes.c_longlong(0), ctypes.c_longlong(0)
        L.check(L.lib().hyp_sys_search_screen_stats(self._h, ctypes.byref(u), ctypes.byref(a), ctypes.byref(b)), "hyp_sys_search_screen_stats")
        if not hasattr(self, "_screen_usable"):
            self._screen_usable = bool(u.value)
        return a.value, b.value

    # ---- common.jl:15-76 on the device
    def get_directions_native(self, solver, dir, rhs, min_impr_tol=0.5):
        res_norm, ns = ctypes.c_double(0.0), c_int(0)
        L.check(L.lib().hyp_sys_get_directions(self._h, L.vec_ptr(dir.vec), L.vec_ptr(rhs.vec), float(solver.mu), float(solver.point.tau),
                                               int(solver.max_ref_steps), float(solver.res_norm_cutoff), float(min_impr_tol),
                                               ctypes.byref(res_norm), ctypes.byref(ns)), "hyp_sys_get_directions")
        return res_norm.value, ns.value

    # ---- qrchol.jl:201-257
    def update_lhs_fact(self, solver):
        nc = len(solver.model.cones)
        flags = (c_int * max(nc, 1))()
        info, fb = c_int(0), c_int(0)
        t0 = time.perf_counter()
        L.check(L.lib().hyp_sys_update_lhs_fact(self._h, flags, ctypes.byref(info), ctypes.byref(fb)), "hyp_sys_update_lhs_fact")
        solver.time_upfact += time.perf_counter() - t0   # (assembly + factor; the split is in hyp_get_timers)
        self.use_sqrt_hess_cones = [bool(flags[k]) for k in range(nc)]
        self.last_info, self.used_fallback = info.value, bool(fb.value)
        self.fallback_kind = fb.value   # 0 Cholesky, 1 Bunch-Kaufman, 2 diagonal shift + Bunch-Kaufman
        if info.value != 0:
            print("positive definite linear system factorization failed")

    # ---- qrchol.jl:39-85
    def solve_subsystem3(self, solver, sol, rhs):
        L.check(L.lib().hyp_sys_solve3(self._h, L.vec_ptr(sol.vec), L.vec_ptr(rhs.vec)), "hyp_sys_solve3")
        return sol

    # ---- qrchol.jl:16-37
    def setup_rhs3(self, model, rhs, sol, rhs_sub):
        for k, cone_k in enumerate(model.cones):
            rhs_z_k = rhs.z_views[k]
            rhs_s_k = rhs.s_views[k]
            rhs_sub_z_k = rhs_sub.z_views[k]
            if cone_k.use_dual_barrier():
                z_temp_k = sol.z_views[k]
                z_temp_k[:] = -rhs_z_k - rhs_s_k
                cone_k.inv_hess_prod(rhs_sub_z_k, z_temp_k)
            else:
                cone_k.hess_prod(rhs_sub_z_k, rhs_z_k)
                rhs_sub_z_k[:] = -rhs_s_k - rhs_sub_z_k

    # ---- common.jl:129-182
    def solve_system(self, solver, sol, rhs):
        model = solver.model
        self.solve_subsystem4(solver, sol, rhs)
        tau = sol.tau
        sol.s[:] = model.h * tau - rhs.z
        self.mul_G(False, sol.x, alpha=-1.0, beta=1.0, y=sol.s)
        taubar = solver.point.tau
        sol.kap = -solver.mu / taubar / taubar * tau + rhs.kap
        return sol

    def solve_subsystem4(self, solver, sol, rhs):
        model = solver.model
        rhs_sub, sol_sub = self.rhs_sub, self.sol_sub
        rhs_sub.x[:] = rhs.x
        rhs_sub.y[:] = -rhs.y
        self.setup_rhs3(model, rhs, sol, rhs_sub)
        self.solve_subsystem3(solver, sol_sub, rhs_sub)
        sol_const = self.sol_const
        tau_num = rhs.tau + rhs.kap + dot_obj(model, sol_sub)
        taubar = solver.point.tau
        tau_denom = solver.mu / taubar / taubar - dot_obj(model, sol_const)
        sol_tau = tau_num / tau_denom
        dim3 = sol_sub.vec.shape[0]
        sol.vec[:dim3] = sol_sub.vec + sol_tau * sol_const.vec
        sol.tau = sol_tau
        return sol

    def get_lhs(self):
        nmp = self.n - self.p
        out = np.zeros((nmp, nmp), order="F")
        L.check(L.lib().hyp_sys_get_lhs(self._h, out.ctypes.data_as(c_vp)), "hyp_sys_get_lhs")
        return out


class SymIndefDenseSystemSolver:
    """SymIndefDenseSystemSolver on the MI355X (symindef.jl:1-56, 203-271): the 3x3 symmetric indefinite form of the Newton
    system, Bunch-Kaufman (rook) on the device.  `load`, `update_lhs`, `solve_subsystem3` forward to hyp_symindef_*; the
    right-hand side set-up and the shared 6 -> 4 -> 3 reductions stay on the host as in the reference.  Use with
    `Solver(reduce=False, ...)` (the reference's option sets for it, test/runnativetests.jl:80-86, 101-118)."""

    native_directions = False     # directions / line search are composed on the host from the per-call entry points

    def __init__(self):
        self._h = None

    def __del__(self):
        try:
            if self._h is not None and L._lib is not None:
                L._lib.hyp_symindef_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def load(self, solver):   # :222-240
        model = solver.model
        n, p, q = model.n, model.p, model.q
        lib = L.lib()
        if self._h is not None:
            lib.hyp_symindef_destroy(self._h)
            self._h = None
        handles = (c_vp * len(model.cones))(*[cone._h for cone in model.cones])
        h = c_vp()
        L.check(lib.hyp_symindef_create(L.ctx(), n, p, q, handles, len(model.cones), ctypes.byref(h)), "hyp_symindef_create")
        self._h = h
        self.n, self.p, self.q = n, p, q
        A = np.asfortranarray(model.A, dtype=np.float64) if p > 0 else None
        G = np.asfortranarray(model.G, dtype=np.float64)
        L.check(lib.hyp_symindef_load(h, A.ctypes.data_as(c_vp) if A is not None else None, G.ctypes.data_as(c_vp)), "hyp_symindef_load")
        # setup_point_sub (common.jl:184-208)
        self.sol_sub = SubPoint(model)
        self.rhs_sub = SubPoint(model)
        self.rhs_const = SubPoint(model)
        self.sol_const = SubPoint(model)
        self.rhs_const.x[:] = -model.c
        self.rhs_const.y[:] = model.b
        self.rhs_const.z[:] = model.h
        self.last_info, self.used_fallback = 0, False
        return self

    def mul_G(self, trans, x, alpha=1.0, beta=0.0, y=None):
        xx = np.ascontiguousarray(x, dtype=np.float64)
        ny = self.n if trans else self.q
        if y is None:
            y = np.zeros(ny)
        L.check(L.lib().hyp_symindef_mul_G(self._h, int(trans), float(alpha), L.vec_ptr(xx), float(beta), L.vec_ptr(y)), "hyp_symindef_mul_G")
        return y

    def update_lhs(self, solver):   # :242-262
        info, fb = c_int(0), c_int(0)
        t0 = time.perf_counter()
        L.check(L.lib().hyp_symindef_update_lhs(self._h, ctypes.byref(info), ctypes.byref(fb)), "hyp_symindef_update_lhs")
        solver.time_upfact += time.perf_counter() - t0
        self.last_info, self.used_fallback = info.value, bool(fb.value)
        if info.value != 0:
            print("symmetric linear system factorization failed")
        self.solve_subsystem3(solver, self.sol_const, self.rhs_const)
        return self

    def solve_subsystem3(self, solver, sol, rhs):   # :264-271
        L.check(L.lib().hyp_symindef_solve3(self._h, L.vec_ptr(sol.vec), L.vec_ptr(rhs.vec)), "hyp_symindef_solve3")
        return sol

    def setup_rhs3(self, model, rhs, sol, rhs_sub):   # :33-56
        for k, cone_k in enumerate(model.cones):
            rhs_z_k = rhs.z_views[k]
            rhs_s_k = rhs.s_views[k]
            rhs_sub_z_k = rhs_sub.z_views[k]
            if cone_k.use_dual_barrier():
                rhs_sub_z_k[:] = -rhs_z_k - rhs_s_k
            else:
                cone_k.inv_hess_prod(rhs_sub_z_k, rhs_s_k)
                rhs_sub_z_k[:] = -rhs_z_k - rhs_sub_z_k

    solve_system = QRCholDenseSystemSolver.solve_system              # common.jl:129-144
    solve_subsystem4 = QRCholDenseSystemSolver.solve_subsystem4      # common.jl:146-182

    def get_lhs(self):
        npq = self.n + self.p + self.q
        out = np.zeros((npq, npq), order="F")
        L.check(L.lib().hyp_symindef_get_lhs(self._h, out.ctypes.data_as(c_vp)), "hyp_symindef_get_lhs")
        return out
