"""Backward error of the blocked Cholesky on a REAL late-iteration Schur matrix (badly scaled, cond ~ 1 / mu^2), in 80-bit
arithmetic: python tools/potrf_accuracy.py make   -> solves psd_blocks(700, [40, 30, 25]) on the GPU up to mu < 1e-6 and saves the
Schur matrix;  python tools/potrf_accuracy.py eval -> factors it under the current HYP_POTRF_* switches and prints
|| U'U - A ||_F / || A ||_F and max_ij |U'U - A|_ij / (|U|'|U|)_ij (componentwise)."""
import ctypes, json, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import hypatia_jl_amd as H
F = "/tmp/schur_late.npy"
if sys.argv[1] == "make":
    from oracle import instances as I
    inst = I.psd_blocks(700, [40, 30, 25], seed=3)
    s = H.Solver(verbose=False)
    keep = {}
    def cb(sv):
        if sv.num_iters >= 1 and sv.mu < 1e-6 and "S" not in keep:
            keep["S"] = sv.syssolver.get_lhs(); keep["mu"] = sv.mu
    s.iter_callback = cb
    s.load(H.make_model(inst)); s.solve()
    S = np.triu(keep["S"]); S = S + np.triu(S, 1).T
    np.save(F, S)
    print("saved Schur matrix at mu = %.2e, cond = %.2e" % (keep["mu"], np.linalg.cond(S)))
else:
    A = np.load(F); n = A.shape[0]
    L = H._lib; lib, ctx = L.lib(), L.ctx()
    Ad = np.asfortranarray(A.copy()); info = ctypes.c_int(-1)
    L.check(lib.hyp_dense_potrf(ctx, n, Ad.ctypes.data_as(ctypes.c_void_p), n, ctypes.byref(info)), "potrf")
    U = np.triu(Ad).astype(np.longdouble)
    R = U.T @ U - A.astype(np.longdouble)
    env = np.abs(U).T @ np.abs(U)
    import scipy.linalg as sla
    Ul = sla.cholesky(A, lower=False).astype(np.longdouble)
    Rl = Ul.T @ Ul - A.astype(np.longdouble)
    print(json.dumps({"info": info.value, "normwise": float(np.linalg.norm(R) / np.linalg.norm(A)), "componentwise": float(np.max(np.abs(R) / env)),
                      "lapack_normwise": float(np.linalg.norm(Rl) / np.linalg.norm(A)), "lapack_componentwise": float(np.max(np.abs(Rl) / (np.abs(Ul).T @ np.abs(Ul))))}))
