"""Cone barrier oracles -- numpy restatement (test infrastructure; see oracle/__init__.py).

Follows /root/reference/src/Cones/:
  Cones.jl                     generic Cone{T} protocol, lazy caches, generic fallbacks
  nonnegative.jl               Nonnegative
  possemideftri.jl             PosSemidefTri (real)
  epinormspectral.jl           EpiNormSpectral (real)
  wsosinterpnonnegative.jl     WSOSInterpNonnegative (real)
Method names mirror the Julia generics; `f!(prod, arr, cone)` becomes `cone.f(prod, arr)` writing
into `prod` in place.  Matrices are handled as 2-D numpy arrays with one column per right-hand
side; vectors as 1-D arrays.
"""
import numpy as np
import scipy.linalg as sla
from scipy.linalg import lapack, blas

from . import arrayutil as au
from . import linalg as la

EPS = np.finfo(np.float64).eps

# Column loops of the PSD products (possemideftri.jl:133-139, 151-156, 168-174, 186-192) are independent per
# column; the reference runs them sequentially with a multi-threaded BLAS.  COLUMN_THREADS > 1 spreads the
# same per-column LAPACK calls over a thread pool -- measured SLOWER here (the per-column numpy glue holds
# the GIL: 0.9 s -> 3.1 s for 1200 columns of side 120 with 8 threads), so it stays at 1 everywhere.
COLUMN_THREADS = 1


def _for_columns(ncols, fn):
    if COLUMN_THREADS <= 1 or ncols < 64:
        for i in range(ncols):
            fn(i)
        return
    from concurrent.futures import ThreadPoolExecutor
    from threadpoolctl import threadpool_limits
    chunk = max(1, ncols // (COLUMN_THREADS * 4))
    def run(lo):
        for i in range(lo, min(ncols, lo + chunk)):
            fn(i)
    with threadpool_limits(limits=1, user_api="blas"):
        with ThreadPoolExecutor(max_workers=COLUMN_THREADS) as ex:
            list(ex.map(run, range(0, ncols, chunk)))


def _cols(arr):
    """view a vector or matrix as (dim, ncols)."""
    return arr.reshape(arr.shape[0], -1)


class Cone:
    """Cones.jl:27-310 (abstract type Cone{T} and its generic methods)."""

    use_dual_barrier_ = False

    # ---- Cones.jl:34-41
    def dimension(self):
        return self.dim

    def get_nu(self):
        return self.nu

    # ---- Cones.jl:138
    def use_dual_barrier(self):
        return self.use_dual_barrier_

    # ---- Cones.jl:126
    def use_dder3(self):
        return True

    # ---- Cones.jl:140-153
    def setup_data(self):
        self.reset_data()
        dim = self.dimension()
        self.point = np.zeros(dim)
        self.dual_point = np.zeros(dim)
        self.grad = np.zeros(dim)
        self.dder3_ = np.zeros(dim)
        self.vec1 = np.zeros(dim)
        self.vec2 = np.zeros(dim)
        self.hess_ = None
        self.inv_hess_ = None
        self.hess_fact = None
        self.setup_extra_data()
        return self

    def setup_extra_data(self):
        pass

    # ---- Cones.jl:157-171
    def load_point(self, point, scal=None):
        if scal is None:
            self.point[:] = point
        else:
            np.multiply(point, scal, out=self.point)

    def load_dual_point(self, point):
        self.dual_point[:] = point

    # ---- Cones.jl:185-186
    def reset_data(self):
        self.feas_updated = self.grad_updated = self.hess_updated = False
        self.inv_hess_updated = self.hess_fact_updated = False

    # ---- Cones.jl:56, 63, 71
    def is_feas(self):
        return self.is_feas_ if self.feas_updated else self.update_feas()

    def is_dual_feas(self):
        return True

    def get_grad(self):
        return self.grad if self.grad_updated else self.update_grad()

    # ---- Cones.jl:79-93
    def hess(self):
        return self.hess_ if self.hess_updated else self.update_hess()

    def inv_hess(self):
        return self.inv_hess_ if self.inv_hess_updated else self.update_inv_hess()

    # ---- Cones.jl:101-105  (mul!(prod, Symmetric(hess,:U), arr))
    def hess_prod(self, prod, arr):
        if not self.hess_updated:
            self.update_hess()
        H = self.hess_
        Hs = np.triu(H) + np.triu(H, 1).T
        _cols(prod)[:] = Hs @ _cols(arr)
        return prod

    # ---- Cones.jl:113-118
    def inv_hess_prod(self, prod, arr):
        self.update_hess_fact()
        _cols(prod)[:] = _cols(self.hess_fact.solve(np.array(_cols(arr), order="F")))
        return prod

    # ---- Cones.jl:189-195
    def use_sqrt_hess_oracles(self, arr_dim):
        if not self.hess_fact_updated:
            if arr_dim < self.dimension():
                return False
            if not self.update_hess_fact():
                return False
        return self.hess_fact.kind == "chol"

    # ---- Cones.jl:198-206  (mul!(prod, hess_fact.U, arr))
    def sqrt_hess_prod(self, prod, arr):
        assert self.hess_fact_updated
        _cols(prod)[:] = blas.dtrmm(1.0, self.hess_fact.factors, np.array(_cols(arr), order="F"),
                                    side=0, lower=0, trans_a=0, diag=0)
        return prod

    # ---- Cones.jl:209-218  (ldiv!(prod, hess_fact.U', arr))
    def inv_sqrt_hess_prod(self, prod, arr):
        assert self.hess_fact_updated
        _cols(prod)[:] = blas.dtrsm(1.0, self.hess_fact.factors, np.array(_cols(arr), order="F"),
                                    side=0, lower=0, trans_a=1, diag=0)
        return prod

    def update_hess_aux(self):
        pass

    # ---- Cones.jl:222-237
    def update_use_hess_prod_slow(self):
        if not self.hess_updated:
            self.update_hess()
        H = self.hess_
        Hs = np.triu(H) + np.triu(H, 1).T
        rel_viol = abs(1 - self.point @ (Hs @ self.point) / self.get_nu())
        self.use_hess_prod_slow = rel_viol > self.dimension() * np.sqrt(EPS)
        self.use_hess_prod_slow_updated = True

    def hess_prod_slow(self, prod, arr):
        return self.hess_prod(prod, arr)

    # ---- Cones.jl:239-251 (posdef_fact_copy!(hess_fact_mat, hess, false): no diagonal shift)
    def update_hess_fact(self):
        if self.hess_fact_updated:
            return True
        if not self.hess_updated:
            self.update_hess()
        self.hess_fact = la.posdef_fact_copy(self.hess_, try_shift=False)
        self.hess_fact_updated = True
        return self.hess_fact.success

    # ---- Cones.jl:253-259
    def update_inv_hess(self):
        self.update_hess_fact()
        assert self.hess_fact.kind == "chol"
        self.inv_hess_ = la.inv_fact_chol(self.hess_fact)
        self.inv_hess_updated = True
        return self.inv_hess_

    # ---- Cones.jl:273-290
    def check_numerics(self, gtol=np.sqrt(np.sqrt(EPS)), Htol=None):
        if Htol is None:
            Htol = 10 * np.sqrt(gtol)
        g = self.get_grad()
        dim = g.shape[0]
        nu = self.get_nu()
        if abs(1 + g @ self.point / nu) > gtol * dim:
            return False
        Hig = self.inv_hess_prod(self.vec1, g)
        if abs(1 - Hig @ g / nu) > Htol * dim:
            return False
        return True

    # ---- Cones.jl:294-310
    def get_proxsqr(self, irtmu, use_max_prox, negtol=np.sqrt(EPS)):
        g = self.get_grad()
        vec1, vec2 = self.vec1, self.vec2
        vec1[:] = irtmu * self.dual_point + g
        self.inv_hess_prod(vec2, vec1)
        prox_sqr = vec2 @ vec1
        if prox_sqr < -negtol * g.shape[0]:
            return np.inf
        return abs(prox_sqr)


# ----------------------------------------------------------------------------------------------
class Nonnegative(Cone):
    """nonnegative.jl:8-145."""

    def __init__(self, dim):
        assert dim >= 1
        self.dim = dim

    def reset_data(self):   # :35-36
        self.feas_updated = self.grad_updated = self.hess_updated = self.inv_hess_updated = False
        self.hess_fact_updated = False

    def use_sqrt_hess_oracles(self, arr_dim):   # :38
        return True

    def get_nu(self):   # :40
        return self.dim

    def set_initial_point(self, arr):   # :42
        arr[:] = 1.0
        return arr

    def update_feas(self):   # :44-49
        assert not self.feas_updated
        self.is_feas_ = bool(np.all(self.point > EPS))
        self.feas_updated = True
        return self.is_feas_

    def is_dual_feas(self):   # :51
        return bool(np.all(self.dual_point > EPS))

    def update_grad(self):   # :53-58
        assert self.is_feas_
        self.grad[:] = -1.0 / self.point
        self.grad_updated = True
        return self.grad

    def update_hess(self):   # :60-69 (Diagonal)
        if not self.grad_updated:
            self.update_grad()
        self.hess_ = np.diag(self.grad ** 2)
        self.hess_updated = True
        return self.hess_

    def update_inv_hess(self):   # :71-80
        assert self.is_feas_
        self.inv_hess_ = np.diag(self.point ** 2)
        self.inv_hess_updated = True
        return self.inv_hess_

    def hess_prod(self, prod, arr):   # :82-90
        _cols(prod)[:] = _cols(arr) / self.point[:, None] / self.point[:, None]
        return prod

    def inv_hess_prod(self, prod, arr):   # :92-100
        _cols(prod)[:] = _cols(arr) * self.point[:, None] * self.point[:, None]
        return prod

    def sqrt_hess_prod(self, prod, arr):   # :102-110
        _cols(prod)[:] = _cols(arr) / self.point[:, None]
        return prod

    def inv_sqrt_hess_prod(self, prod, arr):   # :112-120
        _cols(prod)[:] = _cols(arr) * self.point[:, None]
        return prod

    def dder3(self, dir):   # :122-125
        self.dder3_[:] = (dir / self.point) ** 2 / self.point
        return self.dder3_

    def get_proxsqr(self, irtmu, use_max_prox, negtol=None):   # :137-145
        v = (self.point * self.dual_point * irtmu - 1) ** 2
        return float(np.max(v) if use_max_prox else np.sum(v))


# ----------------------------------------------------------------------------------------------
class PosSemidefTri(Cone):
    """possemideftri.jl:9-207 (real symmetric case, R = Float64)."""

    def __init__(self, dim):
        assert dim >= 1
        self.dim = dim
        self.rt2 = au.RT2
        self.side = au.svec_side(dim)

    def reset_data(self):   # :51-52
        self.feas_updated = self.grad_updated = self.hess_updated = self.inv_hess_updated = False
        self.hess_fact_updated = False

    def use_sqrt_hess_oracles(self, arr_dim):   # :54
        return True

    def setup_extra_data(self):   # :56-65
        s = self.side
        self.mat = np.zeros((s, s), order="F")
        self.mat3 = np.zeros((s, s), order="F")
        self.mat4 = np.zeros((s, s), order="F")
        self.inv_mat = np.zeros((s, s), order="F")
        self.fact_mat = None

    def get_nu(self):   # :67
        return self.side

    def set_initial_point(self, arr):   # :69-78
        arr[:] = 0
        k = 0
        for i in range(1, self.side + 1):
            arr[k] = 1
            k += i + 1
        return arr

    def update_feas(self):   # :80-90
        assert not self.feas_updated
        au.svec_to_smat(self.mat, self.point, self.rt2)
        self.fact_mat = la.chol_upper(self.mat)
        self.is_feas_ = self.fact_mat.success
        self.feas_updated = True
        return self.is_feas_

    def is_dual_feas(self):   # :92-95
        au.svec_to_smat(self.mat3, self.dual_point, self.rt2)
        return la.chol_upper(self.mat3).success

    def update_grad(self):   # :97-107
        assert self.is_feas_
        self.inv_mat[:] = la.inv_fact_chol(self.fact_mat)
        au.smat_to_svec(self.grad, self.inv_mat, self.rt2)
        self.grad *= -1
        au.copytri_upper(self.mat)
        self.grad_updated = True
        return self.grad

    def update_hess(self):   # :109-116
        assert self.grad_updated
        au.copytri_upper(self.inv_mat)
        self.hess_ = np.zeros((self.dim, self.dim))
        au.symm_kron(self.hess_, self.inv_mat, self.rt2)
        self.hess_updated = True
        return self.hess_

    def update_inv_hess(self):   # :118-124
        assert self.is_feas()
        self.inv_hess_ = np.zeros((self.dim, self.dim))
        m = np.triu(self.mat) + np.triu(self.mat, 1).T
        au.symm_kron(self.inv_hess_, m, self.rt2)
        self.inv_hess_updated = True
        return self.inv_hess_

    def _unpack_full(self, col):
        au.svec_to_smat(self.mat4, col, self.rt2)
        au.copytri_upper(self.mat4)
        return self.mat4

    def _unpack_full_new(self, col):
        m = np.zeros((self.side, self.side), order="F")
        au.svec_to_smat(m, col, self.rt2)
        au.copytri_upper(m)
        return m

    def hess_prod(self, prod, arr):   # :126-142   X^-1 V X^-1 via rdiv!/ldiv! with the Cholesky
        assert self.is_feas()
        P, A = _cols(prod), _cols(arr)
        F = self.fact_mat.factors

        def one(i):
            V = self._unpack_full_new(A[:, i])
            # rdiv!(V, fact): V <- V * X^-1 = (X^-1 V')' ; V symmetric on entry
            T = lapack.dpotrs(F, np.asfortranarray(V.T), lower=0)[0].T
            W = lapack.dpotrs(F, np.asfortranarray(T), lower=0)[0]
            au.smat_to_svec(P[:, i], W, self.rt2)
        _for_columns(A.shape[1], one)
        return prod

    def inv_hess_prod(self, prod, arr):   # :144-159   X V X (two symm products)
        assert self.is_feas()
        P, A = _cols(prod), _cols(arr)
        X = np.triu(self.mat) + np.triu(self.mat, 1).T

        def one(i):
            m4 = np.zeros((self.side, self.side), order="F")
            au.svec_to_smat(m4, A[:, i], self.rt2)
            V = np.triu(m4) + np.triu(m4, 1).T
            W = X @ (V @ X)
            au.smat_to_svec(P[:, i], W, self.rt2)
        _for_columns(A.shape[1], one)
        return prod

    def sqrt_hess_prod(self, prod, arr):   # :161-177   U^-T V U^-1
        assert self.is_feas()
        P, A = _cols(prod), _cols(arr)
        F = self.fact_mat.factors

        def one(i):
            V = self._unpack_full_new(A[:, i])
            T = blas.dtrsm(1.0, F, V, side=1, lower=0, trans_a=0, diag=0)   # V U^-1
            W = blas.dtrsm(1.0, F, T, side=0, lower=0, trans_a=1, diag=0)   # U^-T (.)
            au.smat_to_svec(P[:, i], W, self.rt2)
        _for_columns(A.shape[1], one)
        return prod

    def inv_sqrt_hess_prod(self, prod, arr):   # :179-195   U V U'
        assert self.is_feas()
        P, A = _cols(prod), _cols(arr)
        F = self.fact_mat.factors
        for i in range(A.shape[1]):
            V = self._unpack_full(A[:, i])
            T = blas.dtrmm(1.0, F, np.asfortranarray(V), side=1, lower=0, trans_a=1, diag=0)   # V U'
            W = blas.dtrmm(1.0, F, T, side=0, lower=0, trans_a=0, diag=0)                        # U (.)
            au.smat_to_svec(P[:, i], W, self.rt2)
        return prod

    def dder3(self, dir):   # :197-207   X^-1 D X^-1 D X^-1
        assert self.grad_updated
        F = self.fact_mat.factors
        S = self._unpack_full(dir)
        S = lapack.dpotrs(F, np.asfortranarray(S), lower=0)[0]                                  # X^-1 D
        S = blas.dtrsm(1.0, F, np.asfortranarray(S), side=1, lower=0, trans_a=0, diag=0)         # (.) U^-1
        M = S @ S.T
        au.smat_to_svec(self.dder3_, M, self.rt2)
        return self.dder3_


# ----------------------------------------------------------------------------------------------
class EpiNormSpectral(Cone):
    """epinormspectral.jl:13-294 (real case): (u, W) with u >= sigma_1(W), W is d1 x d2, d1 <= d2."""

    def __init__(self, d1, d2, use_dual=False):
        assert 1 <= d1 <= d2
        self.use_dual_barrier_ = use_dual
        self.d1, self.d2 = d1, d2
        self.dim = 1 + d1 * d2

    def reset_data(self):   # :70-72
        self.feas_updated = self.grad_updated = self.hess_updated = self.inv_hess_updated = False
        self.hess_aux_updated = self.hess_fact_updated = False

    def setup_extra_data(self):   # :75-95
        d1, d2 = self.d1, self.d2
        self.W = np.zeros((d1, d2), order="F")
        self.Zi = np.zeros((d1, d1), order="F")
        self.tau = np.zeros((d1, d2), order="F")
        self.HuW = np.zeros((d1, d2), order="F")
        self.WtauI = np.zeros((d2, d2), order="F")
        self.Zitau = np.zeros((d1, d2), order="F")

    def get_nu(self):   # :97
        return self.d1 + 1

    def set_initial_point(self, arr):   # :99-105
        arr[:] = 0
        arr[0] = np.sqrt(self.get_nu())
        return arr

    def _mat(self, v):
        return v.reshape(self.d1, self.d2, order="F")

    def update_feas(self):   # :107-123
        assert not self.feas_updated
        u = self.point[0]
        if u > EPS:
            self.W[:] = self._mat(self.point[1:])
            Z = u * u * np.eye(self.d1) - self.W @ self.W.T
            self.fact_Z = la.chol_upper(Z)
            self.is_feas_ = self.fact_Z.success
        else:
            self.is_feas_ = False
        self.feas_updated = True
        return self.is_feas_

    def is_dual_feas(self):   # :125-132
        u = self.dual_point[0]
        if u > EPS:
            W = self._mat(self.dual_point[1:])
            return bool(u - np.sum(sla.svdvals(W)) > EPS)
        return False

    def update_grad(self):   # :134-150
        assert self.is_feas_
        u = self.point[0]
        self.tau[:] = self.fact_Z.solve(np.asfortranarray(self.W))
        Zi = la.inv_fact_chol(self.fact_Z)
        au.copytri_upper(Zi)
        self.Zi[:] = Zi
        self.grad[0] = -u * np.trace(Zi)
        self.grad[1:] = self.tau.reshape(-1, order="F")
        self.grad *= 2
        self.grad[0] += (self.d1 - 1) / u
        self.grad_updated = True
        return self.grad

    def update_hess_aux(self):   # :152-170
        assert self.grad_updated
        u = self.point[0]
        self.Zitau[:] = self.fact_Z.solve(np.asfortranarray(self.tau))
        self.HuW[:] = -4 * u * self.Zitau
        self.trZi2 = float(np.sum(self.Zi ** 2))
        self.Huu = 4 * u * u * self.trZi2 + (self.grad[0] - 2 * (self.d1 - 1) / u) / u
        self.WtauI[:] = np.eye(self.d2) + self.W.T @ self.tau
        self.hess_aux_updated = True

    def update_hess(self):   # :172-209 (explicit Hessian, upper triangle; vectorized over (l, j) blocks)
        if not self.hess_aux_updated:
            self.update_hess_aux()
        d1, d2 = self.d1, self.d2
        Zi, tau, WtauI = self.Zi, self.tau, self.WtauI
        H = np.zeros((self.dim, self.dim))
        # entry for row (j, i) [W index j + i*d1], column (l, k): Zi[l, j] * WtauI[i, k] + tau[l, i] * tau[j, k]
        # H_WW = kron(WtauI, Zi) + (tau_{l i} tau_{j k}) ; built as a 4-index tensor then reshaped
        T1 = np.einsum("lj,ik->jilk", Zi, WtauI)
        T2 = np.einsum("li,jk->jilk", tau, tau)
        HWW = (T1 + T2).reshape(d1 * d2, d1 * d2, order="F")   # row index j + i*d1, col index l + k*d1
        H[1:, 1:] = 2 * np.triu(HWW)
        H[0, 1:] = self.HuW.reshape(-1, order="F")
        H[0, 0] = self.Huu
        self.hess_ = H
        self.hess_updated = True
        return self.hess_

    def hess_prod(self, prod, arr):   # :211-239
        if not self.hess_aux_updated:
            self.update_hess_aux()
        u = self.point[0]
        W = self.W
        P, A = _cols(prod), _cols(arr)
        for j in range(A.shape[1]):
            a1 = A[0, j]
            AW = self._mat(A[1:, j])
            P[0, j] = self.Huu * a1 + np.sum(self.HuW * AW)
            T = AW @ W.T
            T = T + T.T
            T[np.diag_indices(self.d1)] -= 2 * u * a1
            R = 2 * (T @ self.tau) + 2 * AW
            R = self.fact_Z.solve(np.asfortranarray(R))
            P[1:, j] = R.reshape(-1, order="F")
        return prod

    def dder3(self, dir):   # :241-294
        assert self.hess_aux_updated
        u = self.point[0]
        W = self.W
        u_dir = dir[0]
        W_dir = self._mat(dir[1:]).copy()
        Zi, tau, Zitau, WtauI = self.Zi, self.tau, self.Zitau, self.WtauI
        solve = lambda M: self.fact_Z.solve(np.asfortranarray(M))

        d2d2b = W_dir.T @ tau
        d1d2d = solve(W_dir)
        d1d2b = d1d2d @ WtauI
        d1d2c = d1d2d @ d2d2b.T
        d1d1 = d1d2d @ W.T
        d2d2 = d2d2b @ d2d2b

        d2d2 = d2d2 + W_dir.T @ d1d2b
        d1d2d = tau @ d2d2
        d1d2d = d1d2d + d1d2c @ WtauI
        d1d2d = d1d2d + d1d2b @ d2d2b

        d1d2b = solve(d1d2b)
        d1d2b = d1d2b + Zitau @ d2d2b

        d1d1 = d1d1 + tau @ W_dir.T
        d1d2b = d1d2b + d1d1 @ Zitau
        d1d2b = d1d2b * (-2 * u)

        const1 = 4 * u * u_dir * u
        d1d2c = const1 * Zitau - u_dir * tau
        d1d2c = solve(d1d2c)
        d1d2b = d1d2b + d1d2c

        d1d2d = -2 * u_dir * d1d2b - 2 * d1d2d
        self.dder3_[1:] = d1d2d.reshape(-1, order="F")

        # trZi3 = sum(abs2, ldiv!(tempd1d1, fact_Z.L, Zi))  = || L^-1 Zi ||_F^2 with Z = L L'
        LiZi = sla.solve_triangular(self.fact_Z.factors, Zi, trans="T", lower=False)
        trZi3 = float(np.sum(LiZi ** 2))
        d1d2b = d1d2b + 3 * d1d2c
        self.dder3_[0] = (-np.sum(W_dir * d1d2b) - u * u_dir * (6 * self.trZi2 - 8 * u * trZi3 * u) * u_dir
                          - (self.d1 - 1) * (u_dir / u) ** 2 / u)
        return self.dder3_


# ----------------------------------------------------------------------------------------------
class WSOSInterpNonnegative(Cone):
    """wsosinterpnonnegative.jl:16-200 (real case).  The barrier is for the DUAL cone:
    use_dual_barrier = !use_dual (:58)."""

    def __init__(self, U, Ps, use_dual=False):
        for Pk in Ps:
            assert Pk.shape[0] == U
        self.use_dual_barrier_ = not use_dual
        self.dim = U
        self.Ps = [np.asfortranarray(Pk) for Pk in Ps]
        self.nu = sum(Pk.shape[1] for Pk in Ps)

    def reset_data(self):   # :66-68
        self.feas_updated = self.grad_updated = self.hess_updated = self.inv_hess_updated = False
        self.hess_fact_updated = False
        self.use_hess_prod_slow = self.use_hess_prod_slow_updated = False

    def setup_extra_data(self):   # :70-85
        K = len(self.Ps)
        self.LamF = [None] * K
        self.LamFLP = [None] * K

    def set_initial_point(self, arr):   # :87
        arr[:] = 1.0
        return arr

    def update_feas(self):   # :89-117 (the Ps_order timing sort only changes evaluation order)
        assert not self.feas_updated
        self.is_feas_ = True
        for k, Pk in enumerate(self.Ps):
            LUk = Pk.T * self.point[None, :]
            LLk = LUk @ Pk
            c, info = lapack.dpotrf(LLk, lower=1, clean=0)
            self.LamF[k] = c
            if info != 0:
                self.is_feas_ = False
                break
        self.feas_updated = True
        return self.is_feas_

    def update_grad(self):   # :119-133
        assert self.is_feas_
        self.grad[:] = 0
        for k, Pk in enumerate(self.Ps):
            # LamFLP_k = L_k^-1 P_k'
            LFLP = blas.dtrsm(1.0, self.LamF[k], np.asfortranarray(Pk.T), side=0, lower=1, trans_a=0, diag=0)
            self.LamFLP[k] = LFLP
            self.grad -= np.sum(LFLP ** 2, axis=0)
        self.grad_updated = True
        return self.grad

    def update_hess(self):   # :135-150 (upper triangle)
        assert self.grad_updated
        H = np.zeros((self.dim, self.dim))
        for k in range(len(self.Ps)):
            UU = blas.dsyrk(1.0, self.LamFLP[k], trans=1, lower=0)
            H += np.triu(UU) ** 2
        self.hess_ = H
        self.hess_updated = True
        return self.hess_

    def _partial_lambda(self, dir, LFLP):   # :186-200
        LU = LFLP * dir[None, :]
        LL = LU @ LFLP.T
        LLs = np.triu(LL) + np.triu(LL, 1).T   # Hermitian(LLk) reads the upper triangle
        return LLs @ LFLP

    def hess_prod_slow(self, prod, arr):   # :152-175
        if not self.use_hess_prod_slow_updated:
            self.update_use_hess_prod_slow()
        assert self.hess_updated
        if not self.use_hess_prod_slow:
            return self.hess_prod(prod, arr)
        assert self.grad_updated
        P, A = _cols(prod), _cols(arr)
        P[:] = 0
        for k in range(len(self.Ps)):
            LFLP = self.LamFLP[k]
            for j in range(A.shape[1]):
                LU = self._partial_lambda(A[:, j], LFLP)
                P[:, j] += np.sum(LFLP * LU, axis=0)
        return prod

    def dder3(self, dir):   # :177-188
        assert self.grad_updated
        self.dder3_[:] = 0
        for k in range(len(self.Ps)):
            LU = self._partial_lambda(dir, self.LamFLP[k])
            self.dder3_ += np.sum(LU ** 2, axis=0)
        return self.dder3_


# ----------------------------------------------------------------------------------------------
class LinMatrixIneq(Cone):
    """linmatrixineq.jl:9-159, real dense symmetric `As` (the reference also accepts sparse, Diagonal,
    UniformScaling and complex Hermitian members: converted to dense real by the caller / out of scope).
    Barrier -logdet(sum_i w_i A_i); the explicit Hessian and the generic fallbacks of Cones.jl do the rest."""

    def __init__(self, As, use_dual=False):
        As = [np.array(A, dtype=np.float64) for A in As]
        self.dim = len(As)
        assert self.dim > 1                                    # :42
        self.side = As[0].shape[0]
        for A in As:
            assert A.shape == (self.side, self.side) and np.array_equal(A, A.T)   # :44-53
        assert self.side * (self.side + 1) // 2 >= self.dim    # :56
        assert np.all(np.linalg.eigvalsh(As[0]) > 0)           # :57 isposdef(first(As))
        self.use_dual_barrier_ = bool(use_dual)
        self.As = As
        self.nu = self.side                                    # :72

    def reset_data(self):   # :68-70
        self.feas_updated = self.grad_updated = self.hess_updated = self.inv_hess_updated = False
        self.hess_fact_updated = False
        self.use_hess_prod_slow = self.use_hess_prod_slow_updated = False

    def set_initial_point(self, arr):   # :74-81
        arr[:] = 0.0
        arr[0] = 1.0
        return arr

    def update_feas(self):   # :87-96
        assert not self.feas_updated
        sumA = sum(w * A for w, A in zip(self.point, self.As))
        c, info = lapack.dpotrf(np.asfortranarray(sumA), lower=1, clean=1)
        self.fact_L = c
        self.is_feas_ = (info == 0)
        self.feas_updated = True
        return self.is_feas_

    def update_grad(self):   # :98-109
        assert self.is_feas_
        Lf = self.fact_L
        self.sumAinvAs = []
        for i, A in enumerate(self.As):
            T = blas.dtrsm(1.0, Lf, np.asfortranarray(A), side=0, lower=1, trans_a=0, diag=0)        # L \ A_i
            M = blas.dtrsm(1.0, Lf, np.asfortranarray(T.T), side=0, lower=1, trans_a=0, diag=0)      # L \ (L \ A_i)'
            M = np.triu(M) + np.triu(M, 1).T                                                          # Hermitian(., :U)
            self.sumAinvAs.append(M)
            self.grad[i] = -np.trace(M)
        self.grad_updated = True
        return self.grad

    def update_hess(self):   # :111-123 (upper triangle)
        assert self.grad_updated
        H = np.zeros((self.dim, self.dim))
        for i in range(self.dim):
            for j in range(i, self.dim):
                H[i, j] = np.sum(self.sumAinvAs[i] * self.sumAinvAs[j])
        self.hess_ = H
        self.hess_updated = True
        return self.hess_

    def hess_prod_slow(self, prod, arr):   # :125-144
        if not self.use_hess_prod_slow_updated:
            self.update_use_hess_prod_slow()
        assert self.hess_updated
        if not self.use_hess_prod_slow:
            return self.hess_prod(prod, arr)
        assert self.grad_updated
        P, A = _cols(prod), _cols(arr)
        for j in range(A.shape[1]):
            j_mat = sum(A[i, j] * self.sumAinvAs[i] for i in range(self.dim))
            for i in range(self.dim):
                P[i, j] = np.sum(j_mat * self.sumAinvAs[i])
        return prod

    def dder3(self, dir):   # :146-159
        assert self.grad_updated
        dir_mat = sum(d * M for d, M in zip(dir, self.sumAinvAs))
        Z = dir_mat @ dir_mat.T
        for i in range(self.dim):
            self.dder3_[i] = np.sum(Z * self.sumAinvAs[i])
        return self.dder3_


# ----------------------------------------------------------------------------------------------
class DoublyNonnegativeTri(PosSemidefTri):
    """doublynonnegativetri.jl:9-205: positive semidefinite AND entrywise nonnegative, svec format; barrier
    -logdet(smat(s)) - sum_offdiag log(s_ij).  The PSD oracles plus a diagonal term on the off-diagonal entries; the
    inverse Hessian has no closed form, so it goes through the explicit Hessian and the generic fallbacks."""

    def __init__(self, dim, use_dual=False):
        PosSemidefTri.__init__(self, dim)
        self.use_dual_barrier_ = bool(use_dual)
        # :45-46 (0-based): the svec positions of the strictly-upper entries
        self.offdiag_idxs = np.array([i * (i + 1) // 2 + k for i in range(1, self.side) for k in range(i)], dtype=np.int64)

    def use_sqrt_hess_oracles(self, arr_dim):   # generic rule (Cones.jl:189-195), not the PSD shortcut
        return Cone.use_sqrt_hess_oracles(self, arr_dim)

    def get_nu(self):   # :69
        return self.dim

    def is_dual_feas(self):   # generic default (Cones.jl:59)
        return True

    def set_initial_point(self, arr):   # :71-128
        side, n, d = self.side, float(self.side), float(self.dim)
        if side == 1:
            on_diag = off_diag = 1.0
        elif side == 2:
            on_diag, off_diag = np.sqrt(5.0) / 2, 1 / self.rt2
        else:
            p1 = [-n - 1, 0, n ** 2 + n + 7, 0, -2 * n ** 2 - 8, 0, n ** 2]     # increasing powers (PolynomialRoots.roots)
            on_diag, off_diag = n + 1, 1.0
            for r in np.roots(p1[::-1]):
                offd = float(np.real(r))
                if offd > 0:
                    temp = d - (d - n) * offd ** 2
                    if temp > np.sqrt(EPS):
                        ond = np.sqrt(temp / n)
                        denom = ond ** 2 + (n - 2) / self.rt2 * ond * offd - (n - 1) * offd ** 2 / 2
                        if np.isclose(ond * self.rt2 + (n - 2) * offd, ond * denom * self.rt2, rtol=np.sqrt(EPS), atol=0) and \
                                np.isclose(denom, offd ** 2 * (denom + 1), rtol=np.sqrt(EPS), atol=0):
                            on_diag, off_diag = ond, offd
                            break
        arr[:] = off_diag
        k = 0
        for i in range(1, side + 1):
            arr[k] = on_diag
            k += i + 1
        return arr

    def update_feas(self):   # :130-143
        assert not self.feas_updated
        if np.all(self.point > EPS):
            au.svec_to_smat(self.mat, self.point, self.rt2)
            self.fact_mat = la.chol_upper(self.mat)
            self.is_feas_ = self.fact_mat.success
        else:
            self.is_feas_ = False
        self.feas_updated = True
        return self.is_feas_

    def update_grad(self):   # :145-156
        PosSemidefTri.update_grad(self)
        self.inv_vec = 1.0 / self.point[self.offdiag_idxs]
        self.grad[self.offdiag_idxs] -= self.inv_vec
        return self.grad

    def update_hess(self):   # :158-171
        PosSemidefTri.update_hess(self)
        od = self.offdiag_idxs
        self.hess_[od, od] += self.inv_vec ** 2
        return self.hess_

    def update_inv_hess(self):   # generic (Cones.jl:253-259)
        return Cone.update_inv_hess(self)

    def hess_prod(self, prod, arr):   # :173-192
        PosSemidefTri.hess_prod(self, prod, arr)
        P, A = _cols(prod), _cols(arr)
        od = self.offdiag_idxs
        s_off = self.point[od]
        P[od, :] += A[od, :] / s_off[:, None] / s_off[:, None]
        return prod

    def inv_hess_prod(self, prod, arr):   # generic (Cones.jl:113-118)
        return Cone.inv_hess_prod(self, prod, arr)

    def sqrt_hess_prod(self, prod, arr):   # generic (Cones.jl:198-206)
        return Cone.sqrt_hess_prod(self, prod, arr)

    def inv_sqrt_hess_prod(self, prod, arr):   # generic (Cones.jl:209-218)
        return Cone.inv_sqrt_hess_prod(self, prod, arr)

    def dder3(self, dir):   # :194-205
        PosSemidefTri.dder3(self, dir)
        od = self.offdiag_idxs
        s_off = self.point[od]
        self.dder3_[od] += (dir[od] / s_off) ** 2 / s_off
        return self.dder3_


# ----------------------------------------------------------------------------------------------
class HypoRootdetTri(Cone):
    """hyporootdettri.jl:9-324 (real symmetric case): (u, w) with u <= det(smat(w))^(1/d); barrier
    -log(rootdet(W) - u) - logdet(W).  Closed-form Hessian and inverse-Hessian products."""

    def __init__(self, dim, use_dual=False):
        assert dim >= 2
        self.use_dual_barrier_ = bool(use_dual)
        self.dim = dim
        self.rt2 = au.RT2
        self.d = self._side_of(dim - 1)
        self.di = 1.0 / self.d

    def reset_data(self):   # :61-62
        self.feas_updated = self.grad_updated = self.hess_updated = self.inv_hess_updated = False
        self.hess_fact_updated = False

    def setup_extra_data(self):   # :64-78
        d = self.d
        self.mat = np.zeros((d, d), order="F")
        self.Wi = np.zeros((d, d), order="F")
        self.Wi_vec = np.zeros(self.dim - 1)

    def get_nu(self):   # :80
        return 1 + self.d

    def set_initial_point(self, arr):   # :82-99
        d = self.d
        arr[:] = 0
        c1 = np.sqrt(5.0 * d * d + 2 * d + 1)
        c2 = arr[0] = -np.sqrt((3 * d + 1 - c1) / (2.0 * d + 2))
        c3 = -c2 * (d + 1 + c1) / (2 * d)
        k = 1
        for i in range(1, d + 1):
            arr[k] = c3
            k += self._diag_step(i)
        return arr

    # ---- representation hooks (real symmetric here; oracle/cones_complex.py overrides them for the Hermitian variant)
    def _side_of(self, length):
        return au.svec_side(length)

    def _diag_step(self, i):   # distance from the i-th diagonal entry (1-based) to the next one in the svec
        return i + 1

    def _smat_full(self, v):
        m = np.zeros((self.d, self.d), order="F")
        au.svec_to_smat(m, v, self.rt2)
        au.copytri_upper(m)
        return m

    @staticmethod
    def _full(upper):
        m = np.array(upper, order="F")
        au.copytri_upper(m)
        return m

    def _to_svec(self, out, mat):
        return au.smat_to_svec(out, mat, self.rt2)

    def _kron(self, out, mat):
        return au.symm_kron(out, mat, self.rt2)

    def _chol(self, v):
        """Cholesky of smat(v): (factor object or None, logdet)"""
        m = np.zeros((self.d, self.d), order="F")
        au.svec_to_smat(m, v, self.rt2)
        f = la.chol_upper(m)
        if not f.success:
            return None, 0.0
        return f, 2 * np.sum(np.log(np.diag(f.factors)))

    def _inv_from_chol(self, f):
        return la.inv_fact_chol(f)

    @staticmethod
    def _tr(m):
        return np.trace(m)

    @staticmethod
    def _fro2(m):
        return np.sum(m ** 2)

    def update_feas(self):   # :101-115
        assert not self.feas_updated
        self.fact_W, logdet = self._chol(self.point[1:])
        if self.fact_W is not None:
            self.phi = np.exp(logdet / self.d)
            self.zeta = self.phi - self.point[0]
            self.is_feas_ = self.zeta > EPS
        else:
            self.is_feas_ = False
        self.feas_updated = True
        return self.is_feas_

    def is_dual_feas(self):   # :117-127
        u = self.dual_point[0]
        if u < -EPS:
            f, logdet = self._chol(self.dual_point[1:])
            if f is not None:
                return logdet - self.d * np.log(-u / self.d) > EPS
        return False

    def update_grad(self):   # :129-141
        assert self.is_feas_
        self.phizidi = self.phi / self.zeta * self.di
        self.grad[0] = 1.0 / self.zeta
        self.Wi = self._inv_from_chol(self.fact_W)
        self._to_svec(self.Wi_vec, self.Wi)
        self.grad[1:] = (-self.phizidi - 1) * self.Wi_vec
        self.grad_updated = True
        return self.grad

    def update_hess(self):   # :143-170 (upper triangle)
        assert self.grad_updated
        H = np.zeros((self.dim, self.dim))
        Wi_vec, zeta, pz = self.Wi_vec, self.zeta, self.phizidi
        c1 = -pz / zeta
        c2 = pz * (pz - self.di)
        H[0, 0] = zeta ** -2
        H[0, 1:] = c1 * Wi_vec
        K = np.zeros((self.dim - 1, self.dim - 1))
        self._kron(K, self._full(self.Wi))
        H[1:, 1:] = np.triu((pz + 1) * K + c2 * np.outer(Wi_vec, Wi_vec))
        self.hess_ = H
        self.hess_updated = True
        return self.hess_

    def _two_sided_chol(self, R):
        """U'^-1 R U^-1 (rdiv!(R, U); ldiv!(U', R))"""
        F = self.fact_W.factors
        T = blas.dtrsm(1.0, F, np.asfortranarray(R), side=1, lower=0, trans_a=0, diag=0)
        return blas.dtrsm(1.0, F, T, side=0, lower=0, trans_a=1, diag=0)

    def _two_sided_chol_back(self, S):
        """U^-1 S U'^-1 (rdiv!(S, U'); ldiv!(U, S))"""
        F = self.fact_W.factors
        T = blas.dtrsm(1.0, F, np.asfortranarray(S), side=1, lower=0, trans_a=1, diag=0)
        return blas.dtrsm(1.0, F, T, side=0, lower=0, trans_a=0, diag=0)

    def hess_prod(self, prod, arr):   # :172-203
        assert self.grad_updated
        P, A = _cols(prod), _cols(arr)
        di, zeta, pz = self.di, self.zeta, self.phizidi
        for j in range(A.shape[1]):
            p = A[0, j]
            S = self._two_sided_chol(self._smat_full(A[1:, j]))
            c0 = pz * self._tr(S)
            c1 = c0 - p / zeta
            c2 = pz * c1 - di * c0
            S = (pz + 1) * S
            S[np.diag_indices(self.d)] += c2
            W = self._two_sided_chol_back(S)
            P[0, j] = c1 / -zeta
            self._to_svec(P[1:, j], W)
        return prod

    def update_inv_hess(self):   # :205-233
        assert self.grad_updated
        w = self.point[1:]
        W = self._smat_full(w)
        zeta, phi, di = self.zeta, self.phi, self.di
        phidi = phi * di
        c2 = 1.0 / (self.phizidi + 1)
        c3 = phidi * c2 / zeta * di
        Hi = np.zeros((self.dim, self.dim))
        Hi[0, 0] = zeta ** 2 + phidi * phi
        Hi[0, 1:] = phidi * w
        K = np.zeros((self.dim - 1, self.dim - 1))
        self._kron(K, W)
        Hi[1:, 1:] = np.triu(c2 * K + c3 * np.outer(w, w))
        self.inv_hess_ = Hi
        self.inv_hess_updated = True
        return self.inv_hess_

    def inv_hess_prod(self, prod, arr):   # :235-272
        assert self.grad_updated
        P, A = _cols(prod), _cols(arr)
        w = self.point[1:]
        W = self._smat_full(w)
        zeta, phi, di = self.zeta, self.phi, self.di
        phidi = phi * di
        c2 = 1.0 / (self.phizidi + 1)
        c3 = c2 / zeta * di
        c4 = zeta ** 2 + phidi * phi
        for j in range(A.shape[1]):
            p = A[0, j]
            r = A[1:, j].copy()
            R = self._smat_full(r)
            c5 = w @ r
            c6 = phidi * (c3 * c5 + p)
            P[0, j] = phidi * c5 + c4 * p
            M = W @ (R @ W)
            pw = np.zeros(self.dim - 1)
            self._to_svec(pw, M)
            P[1:, j] = c6 * w + c2 * pw
        return prod

    def dder3(self, dir):   # :274-324
        assert self.grad_updated
        p, r = dir[0], dir[1:]
        zeta, phi, di, pz = self.zeta, self.phi, self.di, self.phizidi
        rwi = self._two_sided_chol(self._smat_full(r))
        c0 = self._tr(rwi) * di
        c6 = self._fro2(rwi) * di
        zichi = (p - phi * c0) / zeta
        c1 = zichi ** 2 + phi / zeta * (c6 - c0 ** 2) / 2
        c7 = pz * (c1 - c6 / 2 + c0 * (zichi + c0 / 2))
        c8 = -pz * (zichi + c0)
        c9 = pz + 1
        self.dder3_[0] = c1 / -zeta
        aux2 = c9 * rwi + c8 * np.eye(self.d)
        M = rwi @ aux2
        M[np.diag_indices(self.d)] += c7
        self._to_svec(self.dder3_[1:], self._two_sided_chol_back(M))
        return self.dder3_


# ----------------------------------------------------------------------------------------------
# hypoperlog.jl:289-319: central ray (u, v, w) of the hypograph-of-perspective-of-sum-log cone, shared by HypoPerLogdetTri
_CENTRAL_RAYS_HYPOPERLOG = np.array([
    [-0.827838387, 0.805102007, 1.290927686], [-0.689607388, 0.724605082, 1.224617936], [-0.584372665, 0.68128058, 1.182421942],
    [-0.503499342, 0.65448622, 1.153053152], [-0.440285893, 0.636444224, 1.131466926], [-0.389979809, 0.623569352, 1.114979519],
    [-0.349255921, 0.613978276, 1.102013921], [-0.315769104, 0.606589839, 1.091577908], [-0.287837744, 0.600745284, 1.083013],
    [-0.264242734, 0.596019009, 1.075868782]])


def get_central_ray_hypoperlog(d):
    if d <= 10:
        return _CENTRAL_RAYS_HYPOPERLOG[d - 1]
    x = 1.0 / d
    if d <= 70:
        return np.array([4.657876 * x ** 2 - 3.116192 * x + 0.000647, 0.424682 * x + 0.553392, 0.760412 * x + 1.001795])
    return np.array([-3.011166 * x - 0.000122, 0.395308 * x + 0.553955, 0.837545 * x + 1.000024])


class HypoPerLogdetTri(HypoRootdetTri):
    """hypoperlogdettri.jl:9-368 (real symmetric case): (u, v, w) with u <= v logdet(smat(w) / v); barrier
    -log(v logdet(W / v) - u) - log(v) - logdet(W).  (Subclass only to share the small matrix helpers.)"""

    def __init__(self, dim, use_dual=False):
        assert dim >= 3
        self.use_dual_barrier_ = bool(use_dual)
        self.dim = dim
        self.rt2 = au.RT2
        self.d = self._side_of(dim - 2)

    def setup_extra_data(self):   # :62-76
        d = self.d
        self.mat = np.zeros((d, d), order="F")
        self.Wi = np.zeros((d, d), order="F")
        self.Wi_vec = np.zeros(self.dim - 2)

    def get_nu(self):   # :78
        return 2 + self.d

    def set_initial_point(self, arr):   # :80-95
        arr[:] = 0
        arr[0], arr[1], w = get_central_ray_hypoperlog(self.d)
        k = 2
        for i in range(1, self.d + 1):
            arr[k] = w
            k += self._diag_step(i)
        return arr

    def update_feas(self):   # :97-118
        assert not self.feas_updated
        v = self.point[1]
        self.is_feas_ = False
        if v > EPS:
            u = self.point[0]
            self.fact_W, logdet = self._chol(self.point[2:])
            if self.fact_W is not None:
                self.phi = logdet - self.d * np.log(v)
                self.zeta = v * self.phi - u
                self.is_feas_ = self.zeta > EPS
        self.feas_updated = True
        return self.is_feas_

    def is_dual_feas(self):   # :120-131
        u = self.dual_point[0]
        if u < -EPS:
            v = self.dual_point[1]
            f, logdet = self._chol(self.dual_point[2:])
            if f is not None:
                return v - u * (logdet + self.d * (1 - np.log(-u))) > EPS
        return False

    def update_grad(self):   # :133-150
        assert self.is_feas_
        v, zeta = self.point[1], self.zeta
        self.grad[0] = 1.0 / zeta
        self.grad[1] = -1.0 / v - (self.phi - self.d) / zeta
        self.Wi = self._inv_from_chol(self.fact_W)
        self._to_svec(self.Wi_vec, self.Wi)
        self.grad[2:] = (-1 - v / zeta) * self.Wi_vec
        self.grad_updated = True
        return self.grad

    def update_hess(self):   # :152-193 (upper triangle)
        assert self.grad_updated
        v, d, zeta = self.point[1], self.d, self.zeta
        zi = 1.0 / zeta
        sigma = self.phi - d
        Wi_vec = self.Wi_vec
        zisig = sigma / zeta
        vzi = v / zeta
        H = np.zeros((self.dim, self.dim))
        H[0, 0] = zi ** 2
        H[0, 1] = -zi * zisig
        H[1, 1] = v ** -2 + zisig ** 2 + d / (v * zeta)
        H[0, 2:] = (-vzi / zeta) * Wi_vec
        H[1, 2:] = ((sigma * vzi - 1) / zeta) * Wi_vec
        K = np.zeros((self.dim - 2, self.dim - 2))
        self._kron(K, self._full(self.Wi))
        Wv = vzi * Wi_vec
        H[2:, 2:] = np.triu((1 + vzi) * K + np.outer(Wv, Wv))
        self.hess_ = H
        self.hess_updated = True
        return self.hess_

    def hess_prod(self, prod, arr):   # :195-236
        assert self.grad_updated
        P, A = _cols(prod), _cols(arr)
        v, d, zeta = self.point[1], self.d, self.zeta
        sigma = self.phi - d
        vzi1 = v / zeta + 1
        for j in range(A.shape[1]):
            p, q = A[0, j], A[1, j]
            S = self._two_sided_chol(self._smat_full(A[2:, j]))
            qzi = q / zeta
            c0 = self._tr(S) / zeta
            c1 = (v * c0 - p / zeta + sigma * qzi) / zeta
            c3 = c1 * v - qzi
            P[0, j] = -c1
            P[1, j] = c1 * sigma - c0 + (qzi * d + q / v) / v
            S = vzi1 * S
            S[np.diag_indices(d)] += c3
            self._to_svec(P[2:, j], self._two_sided_chol_back(S))
        return prod

    def _inv_consts(self):
        v, d, zeta, phi = self.point[1], self.d, self.zeta, self.phi
        zv = zeta + v
        zzvi = zeta / zv
        c3 = v / (zv + d * v)
        c0 = phi - d * zzvi
        return v, d, zeta, phi, zv, zzvi, c3, c0

    def update_inv_hess(self):   # :238-269
        assert self.grad_updated
        v, d, zeta, phi, zv, zzvi, c3, c0 = self._inv_consts()
        w = self.point[2:]
        W = self._smat_full(w)
        c2 = v * c3
        c4 = c2 * zv
        c1 = v * zzvi + c0 * c2
        Hi = np.zeros((self.dim, self.dim))
        Hi[0, 0] = (v * phi) ** 2 + zeta * (zeta + d * v) - d * (zeta + v * phi) ** 2 * c3
        Hi[0, 1] = c0 * c4
        Hi[1, 1] = c4
        Hi[0, 2:] = c1 * w
        Hi[1, 2:] = c2 * w
        K = np.zeros((self.dim - 2, self.dim - 2))
        self._kron(K, W)
        Hi[2:, 2:] = np.triu(zzvi * K + (c2 / zv) * np.outer(w, w))
        self.inv_hess_ = Hi
        self.inv_hess_updated = True
        return self.inv_hess_

    def inv_hess_prod(self, prod, arr):   # :271-316
        assert self.grad_updated
        P, A = _cols(prod), _cols(arr)
        v, d, zeta, phi, zv, zzvi, c3, c0 = self._inv_consts()
        w = self.point[2:]
        W = self._smat_full(w)
        c4 = v * c3 * zv
        c6 = (v * phi) ** 2 + zeta * (zeta + d * v) - d * (zeta + v * phi) ** 2 * c3
        c7 = c4 * c0
        c8 = c7 + v * zeta
        for j in range(A.shape[1]):
            p, q = A[0, j], A[1, j]
            r = A[2:, j].copy()
            R = self._smat_full(r)
            c1 = (w @ r) / zv
            c5 = c0 * p + q + c1
            c2 = v * (zzvi * p + c3 * c5)
            P[0, j] = c6 * p + c7 * q + c8 * c1
            P[1, j] = c4 * c5
            pw = np.zeros(self.dim - 2)
            self._to_svec(pw, W @ (R @ W))
            P[2:, j] = c2 * w + zzvi * pw
        return prod

    def dder3(self, dir):   # :318-368
        assert self.grad_updated
        v, d, zeta = self.point[1], self.d, self.zeta
        p, q, r = dir[0], dir[1], dir[2:]
        sigma = self.phi - d
        viq = q / v
        viq2 = viq ** 2
        vzi = v / zeta
        vzi1 = vzi + 1
        rwi = self._two_sided_chol(self._smat_full(r))
        c0 = self._tr(rwi)
        c7 = self._fro2(rwi)
        zichi = (-p + sigma * q + c0 * v) / zeta
        c4 = (viq * (-viq * d + 2 * c0) - c7) / zeta / 2
        c1 = (zichi ** 2 - v * c4) / zeta
        c3 = -(zichi + viq) / zeta
        c5 = c3 * q + vzi * viq2
        c6 = -2 * vzi * viq - c3 * v
        c8 = c5 + c1 * v
        self.dder3_[0] = -c1
        self.dder3_[1] = c1 * sigma + (viq2 - (d * c5 + c6 * c0 + vzi * c7)) / v - c4
        aux2 = vzi1 * rwi + c6 * np.eye(d)
        M = rwi @ aux2
        M[np.diag_indices(d)] += c8
        self._to_svec(self.dder3_[2:], self._two_sided_chol_back(M))
        return self.dder3_


# ----------------------------------------------------------------------------------------------
class WSOSInterpPosSemidefTri(Cone):
    """wsosinterppossemideftri.jl:9-321: R x R symmetric matrices of polynomials (U interpolant values each, svec order by
    blocks of length U) that are weighted-SOS positive semidefinite.  The barrier is for the DUAL cone
    (use_dual_barrier = !use_dual, :61): -sum_k logdet Lambda_k, Lambda_k the LR x LR block matrix with blocks
    P_k' diag(s_pq) P_k (off-diagonal blocks scaled by 1/sqrt 2).  Explicit Hessian + the generic fallbacks."""

    def __init__(self, R, U, Ps, use_dual=False):
        for Pk in Ps:
            assert Pk.shape[0] == U
        self.use_dual_barrier_ = not use_dual
        self.R, self.U = R, U
        self.dim = U * (R * (R + 1) // 2)
        self.Ps = [np.asfortranarray(Pk) for Pk in Ps]
        self.nu = R * sum(Pk.shape[1] for Pk in Ps)
        self.rt2 = au.RT2
        self.rt2i = 1.0 / au.RT2

    def reset_data(self):   # :70-73
        self.feas_updated = self.grad_updated = self.hess_updated = self.inv_hess_updated = False
        self.hess_fact_updated = False
        self.use_hess_prod_slow = self.use_hess_prod_slow_updated = False

    def _blk(self, row, col):   # 0-based block of svec_idx(row, col), row >= col
        b = row * (row + 1) // 2 + col
        return slice(self.U * b, self.U * (b + 1))

    def set_initial_point(self, arr):   # :100-108
        arr[:] = 0
        for i in range(self.R):
            arr[self._blk(i, i)] = 1.0
        return arr

    def _block_matrix(self, vec, Pk):
        """LR x LR symmetric matrix with blocks P_k' diag(vec_pq) P_k, off-diagonal blocks scaled by 1/sqrt(2) (:122-130, 300-307)"""
        R, L = self.R, Pk.shape[1]
        M = np.zeros((L * R, L * R))
        for p in range(R):
            for q in range(p + 1):
                t = vec[self._blk(p, q)] * (1.0 if p == q else self.rt2i)
                B = (Pk.T * t[None, :]) @ Pk
                M[L * p:L * (p + 1), L * q:L * (q + 1)] = B
                if p != q:
                    M[L * q:L * (q + 1), L * p:L * (p + 1)] = B.T
        return M

    def update_feas(self):   # :110-140
        assert not self.feas_updated
        self.is_feas_ = True
        self.LamFL = [None] * len(self.Ps)
        for k, Pk in enumerate(self.Ps):
            c, info = lapack.dpotrf(self._block_matrix(self.point, Pk), lower=1, clean=1)
            self.LamFL[k] = c
            if info != 0:
                self.is_feas_ = False
                break
        self.feas_updated = True
        return self.is_feas_

    def _block_diag_prod(self, vect, mat1, mat2):   # :264-286: diagonal of every (i, j) U x U block of mat1' mat2
        U = self.U
        for j in range(self.R):
            for i in range(j + 1):
                d = np.einsum("lu,lu->u", mat1[:, U * i:U * (i + 1)], mat2[:, U * j:U * (j + 1)])
                vect[self._blk(j, i)] += d * (1.0 if i == j else self.rt2)

    def update_grad(self):   # :142-186
        assert self.is_feas_
        self.grad[:] = 0
        self.LamFLP = []
        for k, Pk in enumerate(self.Ps):
            KP = np.kron(np.eye(self.R), Pk.T)                                    # kron(I, P'): LR x UR
            FLP = blas.dtrsm(1.0, self.LamFL[k], np.asfortranarray(KP), side=0, lower=1, trans_a=0, diag=0)
            self.LamFLP.append(FLP)
            self._block_diag_prod(self.grad, FLP, FLP)
        self.grad *= -1
        self.grad_updated = True
        return self.grad

    def update_hess(self):   # :188-236 (upper triangle)
        assert self.grad_updated
        R, U = self.R, self.U
        H = np.zeros((self.dim, self.dim))
        for FLP in self.LamFLP:
            PLiP = FLP.T @ FLP
            blkU = lambda a, b: PLiP[U * a:U * (a + 1), U * b:U * (b + 1)]
            for p in range(R):
                for q in range(p + 1):
                    b1 = p * (p + 1) // 2 + q
                    for p2 in range(R):
                        for q2 in range(p2 + 1):
                            b2 = p2 * (p2 + 1) // 2 + q2
                            if b2 < b1:
                                continue
                            scal = self.rt2 if ((p == q) != (p2 == q2)) else 1.0
                            Hv = blkU(p, p2) * blkU(q, q2) * scal
                            if p != q and p2 != q2:
                                Hv = Hv + blkU(p, q2) * blkU(q, p2)
                            H[U * b1:U * (b1 + 1), U * b2:U * (b2 + 1)] += Hv
        self.hess_ = np.triu(H)
        self.hess_updated = True
        return self.hess_

    def _partial_prod(self, prod, arr, use_symm_prod):   # :288-321
        assert self.grad_updated
        P, A = _cols(prod), _cols(arr)
        P[:] = 0
        for k, Pk in enumerate(self.Ps):
            Lf, FLP = self.LamFL[k], self.LamFLP[k]
            for j in range(A.shape[1]):
                M = self._block_matrix(A[:, j], Pk)
                M = blas.dtrsm(1.0, Lf, np.asfortranarray(M), side=0, lower=1, trans_a=0, diag=0)      # L \ M
                M = blas.dtrsm(1.0, Lf, M, side=1, lower=1, trans_a=1, diag=0)                          # (.) / L'
                M = np.triu(M) + np.triu(M, 1).T                                                        # Symmetric(., :U)
                LRUR = M @ FLP
                self._block_diag_prod(P[:, j], LRUR if use_symm_prod else FLP, LRUR)
        return prod

    def hess_prod_slow(self, prod, arr):   # :238-247
        if not self.use_hess_prod_slow_updated:
            self.update_use_hess_prod_slow()
        assert self.hess_updated
        if not self.use_hess_prod_slow:
            return self.hess_prod(prod, arr)
        return self._partial_prod(prod, arr, False)

    def dder3(self, dir):   # :249-252
        assert self.grad_updated
        self._partial_prod(self.dder3_, dir, True)
        return self.dder3_
