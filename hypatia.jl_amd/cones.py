"""Host-side mirror of the reference's cone interface for the HIP cones.

Each class is the Python counterpart of the Julia glue type a maintainer adds next to
`Cones.PosSemidefTri` etc.: the same generics as /root/reference/src/Cones/Cones.jl
(`load_point`, `is_feas`, `grad`, `hess_prod!`, ... here as methods), forwarding to the device
object through the C-ABI.  Host mirrors of `point`, `dual_point`, `vec1`, `vec2` exist because the
steppers read those fields directly (steppers/common.jl:37-46, 96-105).
"""
import ctypes

import numpy as np

from . import _lib as L

c_int, c_dbl, c_vp = ctypes.c_int, ctypes.c_double, ctypes.c_void_p


class Cone:
    """Cones.jl:27-310 (the protocol); every oracle runs on the device."""

    def __init__(self, handle):
        self._h = handle
        lib = L.lib()
        d = c_int(0)
        L.check(lib.hyp_cone_dimension(self._h, ctypes.byref(d)), "hyp_cone_dimension")
        self.dim = d.value
        nu = c_dbl(0)
        L.check(lib.hyp_cone_get_nu(self._h, ctypes.byref(nu)), "hyp_cone_get_nu")
        self.nu = nu.value
        udb = c_int(0)
        L.check(lib.hyp_cone_use_dual_barrier(self._h, ctypes.byref(udb)), "hyp_cone_use_dual_barrier")
        self._use_dual_barrier = bool(udb.value)
        self.setup_data()

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and L._lib is not None:
                L._lib.hyp_cone_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # Cones.jl:34-41, 126, 138
    def dimension(self):
        return self.dim

    def get_nu(self):
        return self.nu

    def use_dual_barrier(self):
        return self._use_dual_barrier

    def use_dder3(self):
        return True

    # Cones.jl:140-153
    def setup_data(self):
        self.reset_data()
        self.point = np.zeros(self.dim)
        self.dual_point = np.zeros(self.dim)
        self.grad = np.zeros(self.dim)
        self.dder3_ = np.zeros(self.dim)
        self.vec1 = np.zeros(self.dim)
        self.vec2 = np.zeros(self.dim)
        self._grad_host_valid = False
        return self

    def set_initial_point(self, arr):
        tmp = np.zeros(self.dim)
        L.check(L.lib().hyp_cone_set_initial_point(self._h, L.vec_ptr(tmp)), "set_initial_point")
        arr[:] = tmp
        return arr

    # Cones.jl:157-171
    def load_point(self, point, scal=None):
        pt = np.ascontiguousarray(point, dtype=np.float64)
        s = 1.0 if scal is None else float(scal)
        if scal is None:
            self.point[:] = pt
        else:
            np.multiply(pt, s, out=self.point)
        L.check(L.lib().hyp_cone_load_point(self._h, L.vec_ptr(pt), s), "load_point")

    def load_dual_point(self, point):
        pt = np.ascontiguousarray(point, dtype=np.float64)
        self.dual_point[:] = pt
        L.check(L.lib().hyp_cone_load_dual_point(self._h, L.vec_ptr(pt)), "load_dual_point")

    # host mirrors after a device-side load_point / load_dual_point / reset_data (hyp_sys_check_cone_points)
    def _mirror_loaded(self, point, scal, dual_point):
        np.multiply(point, scal, out=self.point)
        self.dual_point[:] = dual_point
        self._grad_host_valid = False
        self._reset_host_flags()

    def _reset_host_flags(self):
        pass

    # Cones.jl:185-186
    def reset_data(self):
        if getattr(self, "_h", None) is not None:
            L.check(L.lib().hyp_cone_reset_data(self._h), "reset_data")
        self._grad_host_valid = False

    def _flag(self, fn, *args):
        out = c_int(0)
        L.check(fn(self._h, *args, ctypes.byref(out)), fn.__name__)
        return bool(out.value)

    # Cones.jl:56-71
    def is_feas(self):
        return self._flag(L.lib().hyp_cone_is_feas)

    def is_dual_feas(self):
        return self._flag(L.lib().hyp_cone_is_dual_feas)

    def get_grad(self):
        if not self._grad_host_valid:
            L.check(L.lib().hyp_cone_grad(self._h, L.vec_ptr(self.grad)), "grad")
            self._grad_host_valid = True
        return self.grad

    def _prod(self, fn, prod, arr):
        pa, pp, ldp, ncp, pcopy = L.mat_view(prod)
        aa, ap, lda, nca, _ = L.mat_view(arr)
        assert ncp == nca and pa.shape[0] == self.dim and aa.shape[0] == self.dim
        L.check(fn(self._h, pp, ldp, ap, lda, nca), fn.__name__)
        if pcopy:
            prod[...] = pa
        return prod

    # Cones.jl:101-118, 198-237
    def hess_prod(self, prod, arr):
        return self._prod(L.lib().hyp_cone_hess_prod, prod, arr)

    def inv_hess_prod(self, prod, arr):
        return self._prod(L.lib().hyp_cone_inv_hess_prod, prod, arr)

    def hess_prod_slow(self, prod, arr):
        return self._prod(L.lib().hyp_cone_hess_prod_slow, prod, arr)

    def use_sqrt_hess_oracles(self, arr_dim):
        return self._flag(L.lib().hyp_cone_use_sqrt_hess_oracles, int(arr_dim))

    def sqrt_hess_prod(self, prod, arr):
        return self._prod(L.lib().hyp_cone_sqrt_hess_prod, prod, arr)

    def inv_sqrt_hess_prod(self, prod, arr):
        return self._prod(L.lib().hyp_cone_inv_sqrt_hess_prod, prod, arr)

    # Cones.jl:134
    def dder3(self, dir):
        d = np.ascontiguousarray(dir, dtype=np.float64)
        L.check(L.lib().hyp_cone_dder3(self._h, L.vec_ptr(d), L.vec_ptr(self.dder3_)), "dder3")
        return self.dder3_

    def update_hess_aux(self):
        pass

    # Cones.jl:273-310
    def check_numerics(self):
        return self._flag(L.lib().hyp_cone_check_numerics)

    def get_proxsqr(self, irtmu, use_max_prox):
        out = c_dbl(0)
        L.check(L.lib().hyp_cone_get_proxsqr(self._h, float(irtmu), int(bool(use_max_prox)), ctypes.byref(out)), "get_proxsqr")
        return out.value

    # Cones.jl:79-93 (explicit, tests only)
    def hess(self):
        H = np.zeros((self.dim, self.dim), order="F")
        L.check(L.lib().hyp_cone_hess(self._h, H.ctypes.data_as(c_vp)), "hess")
        return H

    def inv_hess(self):
        H = np.zeros((self.dim, self.dim), order="F")
        L.check(L.lib().hyp_cone_inv_hess(self._h, H.ctypes.data_as(c_vp)), "inv_hess")
        return H


class Nonnegative(Cone):
    """Cones.Nonnegative{Float64}(dim)  (nonnegative.jl:8-33)."""
    is_nonnegative = True   # process.jl:37 special-cases `cone isa Cones.Nonnegative` in rescale_data

    def __init__(self, dim):
        h = c_vp()
        L.check(L.lib().hyp_cone_create_nonnegative(L.ctx(), int(dim), ctypes.byref(h)), "hyp_cone_create_nonnegative")
        super().__init__(h)


class PosSemidefTri(Cone):
    """Cones.PosSemidefTri{Float64, Float64}(dim)  (possemideftri.jl:9-46)."""

    def __init__(self, dim):
        h = c_vp()
        L.check(L.lib().hyp_cone_create_possemideftri(L.ctx(), int(dim), ctypes.byref(h)), "hyp_cone_create_possemideftri")
        super().__init__(h)
        self.side = int(round((np.sqrt(1 + 8 * dim) - 1) / 2))


class PosSemidefTriComplex(Cone):
    """Cones.PosSemidefTri{Float64, ComplexF64}(dim)  (possemideftri.jl:9-46 with R = Complex{T}; dim = side^2)."""

    def __init__(self, dim):
        h = c_vp()
        L.check(L.lib().hyp_cone_create_possemideftri_complex(L.ctx(), int(dim), ctypes.byref(h)), "hyp_cone_create_possemideftri_complex")
        super().__init__(h)
        self.side = int(round(np.sqrt(dim)))


class _GenericHessMixin:
    """cones that carry Hypatia's `use_hess_prod_slow` switch (Cones.jl:222-237)"""

    @property
    def use_hess_prod_slow(self):
        return self._slow

    @use_hess_prod_slow.setter
    def use_hess_prod_slow(self, v):
        self._slow = bool(v)
        if getattr(self, "_h", None) is not None:   # both values reach the device: host and device never disagree on the path
            L.check(L.lib().hyp_cone_set_use_hess_prod_slow(self._h, int(bool(v))), "set_use_hess_prod_slow")

    def update_use_hess_prod_slow(self):
        out = c_int(0)
        L.check(L.lib().hyp_cone_update_use_hess_prod_slow(self._h, ctypes.byref(out)), "update_use_hess_prod_slow")
        self._slow = bool(out.value)
        self.use_hess_prod_slow_updated = True

    def reset_data(self):
        super().reset_data()
        self._reset_host_flags()

    def _reset_host_flags(self):
        self._slow = False
        self.use_hess_prod_slow_updated = False


class EpiNormSpectral(_GenericHessMixin, Cone):
    """Cones.EpiNormSpectral{Float64, Float64}(d1, d2; use_dual)  (epinormspectral.jl:13-66)."""

    def __init__(self, d1, d2, use_dual=False):
        self._slow = False
        h = c_vp()
        L.check(L.lib().hyp_cone_create_epinormspectral(L.ctx(), int(d1), int(d2), int(bool(use_dual)), ctypes.byref(h)),
                "hyp_cone_create_epinormspectral")
        self.d1, self.d2 = d1, d2
        super().__init__(h)


class EpiNormSpectralComplex(Cone):
    """Cones.EpiNormSpectral{Float64, ComplexF64}(d1, d2; use_dual)  (epinormspectral.jl:13-66 with R = Complex{T}): the cone
    vector is (u, W), W complex d1 x d2 as (re, im) pairs in column-major order; dim = 1 + 2 d1 d2."""

    def __init__(self, d1, d2, use_dual=False):
        h = c_vp()
        L.check(L.lib().hyp_cone_create_epinormspectral_complex(L.ctx(), int(d1), int(d2), int(bool(use_dual)), ctypes.byref(h)),
                "hyp_cone_create_epinormspectral_complex")
        self.d1, self.d2 = d1, d2
        super().__init__(h)


class WSOSInterpNonnegative(_GenericHessMixin, Cone):
    """Cones.WSOSInterpNonnegative{Float64, Float64}(U, Ps; use_dual)  (wsosinterpnonnegative.jl:16-63)."""

    def __init__(self, U, Ps, use_dual=False):
        self._slow = False
        Ps = [np.asfortranarray(P, dtype=np.float64) for P in Ps]
        for P in Ps:
            assert P.shape[0] == U
        K = len(Ps)
        Ls = (c_int * K)(*[P.shape[1] for P in Ps])
        ptrs = (c_vp * K)(*[P.ctypes.data_as(c_vp) for P in Ps])
        h = c_vp()
        L.check(L.lib().hyp_cone_create_wsosinterpnonnegative(L.ctx(), int(U), K, Ls, ptrs, int(bool(use_dual)), ctypes.byref(h)),
                "hyp_cone_create_wsosinterpnonnegative")
        super().__init__(h)


class WSOSInterpNonnegativeComplex(_GenericHessMixin, Cone):
    """Cones.WSOSInterpNonnegative{Float64, ComplexF64}(U, Ps; use_dual)  (wsosinterpnonnegative.jl:15-63 with R = Complex{T}):
    complex bases (PolyUtils/complex.jl:13-72), real cone vector."""

    def __init__(self, U, Ps, use_dual=False):
        self._slow = False
        Ps = [np.asfortranarray(P, dtype=np.complex128) for P in Ps]
        for P in Ps:
            assert P.shape[0] == U                                            # :54-56
        K = len(Ps)
        Ls = (c_int * K)(*[P.shape[1] for P in Ps])
        ptrs = (c_vp * K)(*[P.ctypes.data_as(c_vp) for P in Ps])              # (column-major complex128 = (re, im) interleaved)
        h = c_vp()
        L.check(L.lib().hyp_cone_create_wsosinterpnonnegative_complex(L.ctx(), int(U), K, Ls, ptrs, int(bool(use_dual)), ctypes.byref(h)),
                "hyp_cone_create_wsosinterpnonnegative_complex")
        super().__init__(h)


class LinMatrixIneq(_GenericHessMixin, Cone):
    """Cones.LinMatrixIneq{Float64}(As; use_dual)  (linmatrixineq.jl:9-65), dense real symmetric or complex Hermitian members
    (sparse / Diagonal members are densified here; `I` is not accepted, pass np.eye(side))."""

    def __init__(self, As, use_dual=False):
        self._slow = False
        As = [np.asarray(A.toarray() if hasattr(A, "toarray") else A) for A in As]
        dim, side = len(As), As[0].shape[0]
        assert dim > 1                                                        # :42
        h = c_vp()
        if any(np.iscomplexobj(A) for A in As):   # complex Hermitian members (the cone vector stays real)
            As = [np.asarray(A, dtype=np.complex128) for A in As]
            for A in As:
                assert A.shape == (side, side) and np.array_equal(A, A.conj().T)  # :44-53 ishermitian
            assert side * (side + 1) // 2 >= dim                              # :56
            assert np.all(np.linalg.eigvalsh(As[0]) > 0)                      # :57
            stacked = np.ascontiguousarray(np.stack([A.T for A in As]))       # C order of A.T = column-major A, (re, im) interleaved
            L.check(L.lib().hyp_cone_create_linmatrixineq_complex(L.ctx(), dim, side, stacked.ctypes.data_as(c_vp), int(bool(use_dual)),
                                                                  ctypes.byref(h)), "hyp_cone_create_linmatrixineq_complex")
            super().__init__(h)
            return
        As = [np.asarray(A, dtype=np.float64) for A in As]
        for A in As:
            assert A.shape == (side, side) and np.array_equal(A, A.T)         # :44-53
        assert side * (side + 1) // 2 >= dim                                  # :56
        assert np.all(np.linalg.eigvalsh(As[0]) > 0)                          # :57
        stacked = np.ascontiguousarray(np.stack(As))                          # symmetric: row- and column-major coincide
        L.check(L.lib().hyp_cone_create_linmatrixineq(L.ctx(), dim, side, stacked.ctypes.data_as(c_vp), int(bool(use_dual)), ctypes.byref(h)),
                "hyp_cone_create_linmatrixineq")
        super().__init__(h)


class DoublyNonnegativeTri(_GenericHessMixin, Cone):
    """Cones.DoublyNonnegativeTri{Float64}(dim; use_dual)  (doublynonnegativetri.jl:9-52)."""

    def __init__(self, dim, use_dual=False):
        self._slow = False
        h = c_vp()
        L.check(L.lib().hyp_cone_create_doublynonnegativetri(L.ctx(), int(dim), int(bool(use_dual)), ctypes.byref(h)),
                "hyp_cone_create_doublynonnegativetri")
        super().__init__(h)


class HypoRootdetTri(_GenericHessMixin, Cone):
    """Cones.HypoRootdetTri{Float64, Float64}(dim; use_dual)  (hyporootdettri.jl:9-59)."""

    def __init__(self, dim, use_dual=False):
        self._slow = False
        h = c_vp()
        L.check(L.lib().hyp_cone_create_hyporootdettri(L.ctx(), int(dim), int(bool(use_dual)), ctypes.byref(h)),
                "hyp_cone_create_hyporootdettri")
        super().__init__(h)


class HypoRootdetTriComplex(_GenericHessMixin, Cone):
    """Cones.HypoRootdetTri{Float64, ComplexF64}(dim; use_dual)  (hyporootdettri.jl:9-59 with R = Complex{T}; dim = 1 + side^2)."""

    def __init__(self, dim, use_dual=False):
        self._slow = False
        h = c_vp()
        L.check(L.lib().hyp_cone_create_hyporootdettri_complex(L.ctx(), int(dim), int(bool(use_dual)), ctypes.byref(h)),
                "hyp_cone_create_hyporootdettri_complex")
        super().__init__(h)


class HypoPerLogdetTriComplex(_GenericHessMixin, Cone):
    """Cones.HypoPerLogdetTri{Float64, ComplexF64}(dim; use_dual)  (hypoperlogdettri.jl:9-58 with R = Complex{T}; dim = 2 + side^2)."""

    def __init__(self, dim, use_dual=False):
        self._slow = False
        h = c_vp()
        L.check(L.lib().hyp_cone_create_hypoperlogdettri_complex(L.ctx(), int(dim), int(bool(use_dual)), ctypes.byref(h)),
                "hyp_cone_create_hypoperlogdettri_complex")
        super().__init__(h)


class HypoPerLogdetTri(_GenericHessMixin, Cone):
    """Cones.HypoPerLogdetTri{Float64, Float64}(dim; use_dual)  (hypoperlogdettri.jl:9-58)."""

    def __init__(self, dim, use_dual=False):
        self._slow = False
        h = c_vp()
        L.check(L.lib().hyp_cone_create_hypoperlogdettri(L.ctx(), int(dim), int(bool(use_dual)), ctypes.byref(h)),
                "hyp_cone_create_hypoperlogdettri")
        super().__init__(h)


class WSOSInterpPosSemidefTri(_GenericHessMixin, Cone):
    """Cones.WSOSInterpPosSemidefTri{Float64}(R, U, Ps; use_dual)  (wsosinterppossemideftri.jl:9-69)."""

    def __init__(self, R, U, Ps, use_dual=False):
        self._slow = False
        Ps = [np.asfortranarray(P, dtype=np.float64) for P in Ps]
        for P in Ps:
            assert P.shape[0] == U
        K = len(Ps)
        Ls = (c_int * K)(*[P.shape[1] for P in Ps])
        ptrs = (c_vp * K)(*[P.ctypes.data_as(c_vp) for P in Ps])
        h = c_vp()
        L.check(L.lib().hyp_cone_create_wsosinterppossemideftri(L.ctx(), int(R), int(U), K, Ls, ptrs, int(bool(use_dual)), ctypes.byref(h)),
                "hyp_cone_create_wsosinterppossemideftri")
        super().__init__(h)
