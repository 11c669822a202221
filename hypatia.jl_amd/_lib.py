"""ctypes binding of libhypatia_hip.so (include/hypatia_hip.h).  Loads lazily; fails loudly."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HYP_LIB_PATH") or os.path.join(HERE, "libhypatia_hip.so")   # (HYP_LIB_PATH: another build of the same library, for A/B measurements)

c_int, c_dbl, c_vp = ctypes.c_int, ctypes.c_double, ctypes.c_void_p
P = ctypes.POINTER

_lib = None
_ctx = None


class HypatiaHipError(RuntimeError):
    pass


# name -> argtypes  (every function returns int except hyp_last_error)
SIGNATURES = {
    "hyp_ctx_create": [c_int, P(c_vp)],
    "hyp_ctx_destroy": [c_vp],
    "hyp_ctx_synchronize": [c_vp],
    "hyp_device_count": [P(c_int)],
    "hyp_get_timers": [c_vp, c_vp],
    "hyp_ctx_bk_stats": [c_vp, c_vp],
    "hyp_ctx_plan_stats": [c_vp, c_vp],
    "hyp_reset_timers": [c_vp],
    "hyp_get_kernel_stats": [c_vp, c_vp],
    "hyp_cone_create_nonnegative": [c_vp, c_int, P(c_vp)],
    "hyp_cone_create_possemideftri": [c_vp, c_int, P(c_vp)],
    "hyp_cone_create_possemideftri_complex": [c_vp, c_int, P(c_vp)],
    "hyp_cone_create_epinormspectral": [c_vp, c_int, c_int, c_int, P(c_vp)],
    "hyp_cone_create_epinormspectral_complex": [c_vp, c_int, c_int, c_int, P(c_vp)],
    "hyp_cone_create_wsosinterpnonnegative": [c_vp, c_int, c_int, P(c_int), P(c_vp), c_int, P(c_vp)],
    "hyp_cone_create_wsosinterpnonnegative_complex": [c_vp, c_int, c_int, P(c_int), P(c_vp), c_int, P(c_vp)],
    "hyp_cone_create_wsosinterppossemideftri": [c_vp, c_int, c_int, c_int, P(c_int), P(c_vp), c_int, P(c_vp)],
    "hyp_cone_create_linmatrixineq": [c_vp, c_int, c_int, c_vp, c_int, P(c_vp)],
    "hyp_cone_create_linmatrixineq_complex": [c_vp, c_int, c_int, c_vp, c_int, P(c_vp)],
    "hyp_cone_create_doublynonnegativetri": [c_vp, c_int, c_int, P(c_vp)],
    "hyp_cone_create_hyporootdettri": [c_vp, c_int, c_int, P(c_vp)],
    "hyp_cone_create_hypoperlogdettri": [c_vp, c_int, c_int, P(c_vp)],
    "hyp_cone_create_hyporootdettri_complex": [c_vp, c_int, c_int, P(c_vp)],
    "hyp_cone_create_hypoperlogdettri_complex": [c_vp, c_int, c_int, P(c_vp)],
    "hyp_cone_update_use_hess_prod_slow": [c_vp, P(c_int)],
    "hyp_cone_set_use_hess_prod_slow": [c_vp, c_int],
    "hyp_cone_destroy": [c_vp],
    "hyp_cone_dimension": [c_vp, P(c_int)],
    "hyp_cone_get_nu": [c_vp, P(c_dbl)],
    "hyp_cone_use_dual_barrier": [c_vp, P(c_int)],
    "hyp_cone_set_initial_point": [c_vp, c_vp],
    "hyp_cone_load_point": [c_vp, c_vp, c_dbl],
    "hyp_cone_load_dual_point": [c_vp, c_vp],
    "hyp_cone_reset_data": [c_vp],
    "hyp_cone_get_point": [c_vp, c_vp],
    "hyp_cone_get_dual_point": [c_vp, c_vp],
    "hyp_cone_is_feas": [c_vp, P(c_int)],
    "hyp_cone_is_dual_feas": [c_vp, P(c_int)],
    "hyp_cone_grad": [c_vp, c_vp],
    "hyp_cone_hess_prod": [c_vp, c_vp, c_int, c_vp, c_int, c_int],
    "hyp_cone_inv_hess_prod": [c_vp, c_vp, c_int, c_vp, c_int, c_int],
    "hyp_cone_hess_prod_slow": [c_vp, c_vp, c_int, c_vp, c_int, c_int],
    "hyp_cone_use_sqrt_hess_oracles": [c_vp, c_int, P(c_int)],
    "hyp_cone_sqrt_hess_prod": [c_vp, c_vp, c_int, c_vp, c_int, c_int],
    "hyp_cone_inv_sqrt_hess_prod": [c_vp, c_vp, c_int, c_vp, c_int, c_int],
    "hyp_cone_dder3": [c_vp, c_vp, c_vp],
    "hyp_cone_check_numerics": [c_vp, P(c_int)],
    "hyp_cone_get_proxsqr": [c_vp, c_dbl, c_int, P(c_dbl)],
    "hyp_cone_hess": [c_vp, c_vp],
    "hyp_cone_inv_hess": [c_vp, c_vp],
    "hyp_sys_create": [c_vp, c_int, c_int, c_int, P(c_vp), c_int, P(c_vp)],
    "hyp_sys_destroy": [c_vp],
    "hyp_sys_load": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "hyp_symindef_create": [c_vp, c_int, c_int, c_int, P(c_vp), c_int, P(c_vp)],
    "hyp_symindef_destroy": [c_vp],
    "hyp_symindef_load": [c_vp, c_vp, c_vp],
    "hyp_symindef_update_lhs": [c_vp, P(c_int), P(c_int)],
    "hyp_symindef_solve3": [c_vp, c_vp, c_vp],
    "hyp_symindef_mul_G": [c_vp, c_int, c_dbl, c_vp, c_dbl, c_vp],
    "hyp_symindef_get_lhs": [c_vp, c_vp],
    "hyp_sys_update_lhs_fact": [c_vp, P(c_int), P(c_int), P(c_int)],
    "hyp_sys_solve3": [c_vp, c_vp, c_vp],
    "hyp_sys_get_directions2": [c_vp, c_vp, c_vp, c_dbl, c_dbl, c_int, c_dbl, c_dbl, P(c_dbl), P(c_int)],
    "hyp_sys_step_directions": [c_vp, c_vp, c_vp, c_dbl, c_dbl, c_int, c_dbl, c_dbl, c_vp, c_vp, P(c_int), P(c_int), P(c_int), P(c_int), c_vp],
    "hyp_sys_set_comm": [c_vp, c_vp, c_vp, c_vp, ctypes.c_long],
    "hyp_comm_unique_id": [c_vp],
    "hyp_comm_init_rank": [c_vp, c_int, c_int, c_vp, P(c_vp)],
    "hyp_comm_destroy": [c_vp],
    "hyp_comm_allreduce": [c_vp, c_vp, ctypes.c_long, c_int],
    "hyp_sys_set_comm_rccl": [c_vp, c_vp],
    "hyp_sys_comm_stats": [c_vp, c_vp],
    "hyp_qrcp_factor": [c_vp, c_int, c_int, c_vp, c_int, c_vp, P(c_vp)],
    "hyp_qrcp_get": [c_vp, c_vp, c_vp, c_vp, c_vp],
    "hyp_qrcp_apply_q": [c_vp, c_int, c_vp],
    "hyp_qrcp_destroy": [c_vp],
    "hyp_sys_set_kshard": [c_vp, c_int, c_int],
    "hyp_sys_set_direction_rows": [c_vp, c_int],
    "hyp_sys_last_update_lhs_seconds": [c_vp, P(c_dbl)],
    "hyp_sys_bench_gemv": [c_vp, c_int, c_vp],
    "hyp_sys_search_alpha": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_int, c_dbl, c_dbl, c_int, c_dbl, c_vp,
                             P(c_int), P(c_dbl), P(c_int), P(c_int), P(c_dbl)],
    "hyp_sys_search_screen_stats": [c_vp, P(c_int), c_vp, c_vp],
    "hyp_sys_search_alpha_resident": [c_vp, c_int, c_int, c_vp, c_int, c_int, c_dbl, c_dbl, c_int, c_dbl, c_vp,
                                      P(c_int), P(c_dbl), P(c_int), P(c_int), P(c_dbl)],
    "hyp_sys_check_cone_points": [c_vp, c_vp, c_dbl, c_dbl, c_int, c_dbl, P(c_int), P(c_dbl), P(c_int), P(c_dbl)],
    "hyp_sys_residual_products": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "hyp_sys_residual_products2": [c_vp, c_vp, c_vp, c_vp, c_dbl, c_vp, c_vp, c_vp, c_vp],
    "hyp_sys_set_comm_layout": [c_vp, c_int, c_int],
    "hyp_sys_comm_hist": [c_vp, c_vp],
    "hyp_sys_comm_times": [c_vp, c_vp],
    "hyp_sys_allreduce_host": [c_vp, c_vp, c_int, c_int],
    "hyp_sys_load_model": [c_vp, c_vp, c_vp, c_vp, c_vp],
    "hyp_sys_update_lhs": [c_vp, P(c_int), P(c_int), P(c_int), c_vp],
    "hyp_sys_get_directions": [c_vp, c_vp, c_vp, c_dbl, c_dbl, c_int, c_dbl, c_dbl, P(c_dbl), P(c_int)],
    "hyp_sys_assemble_lhs": [c_vp, P(c_int)],
    "hyp_sys_factor_lhs": [c_vp, P(c_int), P(c_int)],
    "hyp_sys_lhs_export_dev": [c_vp, c_vp],
    "hyp_sys_lhs_import_dev": [c_vp, c_vp],
    "hyp_sys_set_lhs": [c_vp, c_vp],
    "hyp_sys_potrs": [c_vp, c_vp],
    "hyp_sys_block_hess_prod": [c_vp, c_vp, c_vp],
    "hyp_sys_mul_G": [c_vp, c_int, c_dbl, c_vp, c_dbl, c_vp],
    "hyp_sys_get_lhs": [c_vp, c_vp],
    "hyp_dense_gemm": [c_vp, c_int, c_int, c_int, c_int, c_int, c_dbl, c_vp, c_int, c_vp, c_int, c_dbl, c_vp, c_int],
    "hyp_dense_syrk": [c_vp, c_int, c_int, c_vp, c_int, c_vp, c_int],
    "hyp_dense_potrf": [c_vp, c_int, c_vp, c_int, P(c_int)],
    "hyp_dense_posdef_solve": [c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int, P(c_int), P(c_int), P(c_int)],
    "hyp_dense_posv": [c_vp, c_int, c_vp, c_int, c_vp, P(c_int)],
    "hyp_dense_posv_multi": [c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int, P(c_int)],
    "hyp_dense_sysv_rook": [c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int, P(c_int), c_vp, c_vp, c_vp, c_vp],
    "hyp_dense_lstsq_normal": [c_vp, c_int, c_int, c_vp, c_int, c_vp, c_vp, P(c_dbl), P(c_int)],
    "hyp_dense_gemv": [c_vp, c_int, c_int, c_int, c_dbl, c_vp, c_int, c_vp, c_dbl, c_vp],
    "hyp_dense_gemv_both": [c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp, c_dbl, c_vp, c_vp, c_dbl, c_vp, P(c_int)],
    "hyp_bench_syrk": [c_vp, c_int, c_int, c_int, P(c_dbl)],
    "hyp_bench_potrf": [c_vp, c_int, c_int, P(c_dbl)],
    "hyp_bench_trsv": [c_vp, c_int, c_int, P(c_dbl), P(c_dbl)],
}


def load_library():
    """dlopen the library and declare every symbol of include/hypatia_hip.h (no GPU needed)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HypatiaHipError("libhypatia_hip.so is missing at %s: run __graft_entry__.build() "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    lib.hyp_last_error.argtypes = [c_vp]
    lib.hyp_last_error.restype = ctypes.c_char_p
    if os.environ.get("HYP_PROFILE"):
        lib = _ProfiledLib(lib)
    _lib = lib
    return lib


class _ProfiledLib:
    """HYP_PROFILE=1: accumulate wall time and call counts per C-ABI entry point (host-side view)."""

    def __init__(self, lib):
        import time
        self._lib = lib
        self.stats = {}
        self._time = time.perf_counter

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("hyp_") or name == "hyp_last_error":
            return fn
        stats, clock = self.stats, self._time

        def wrapped(*a):
            t0 = clock()
            r = fn(*a)
            st = stats.setdefault(name, [0, 0.0])
            st[0] += 1
            st[1] += clock() - t0
            return r
        wrapped.__name__ = name
        setattr(self, name, wrapped)
        return wrapped

    def report(self):
        rows = sorted(self.stats.items(), key=lambda kv: -kv[1][1])
        return "\n".join("%-36s calls %6d  total %9.3f ms  avg %8.1f us" % (k, v[0], v[1] * 1e3, v[1] / v[0] * 1e6) for k, v in rows)

    def reset(self):
        self.stats.clear()


def lib():
    return load_library()


def ctx():
    """the process-wide context on the local GPU (LOCAL_RANK selects the device)."""
    global _ctx
    if _ctx is None:
        L = load_library()
        n = c_int(0)
        L.hyp_device_count(ctypes.byref(n))
        if n.value <= 0:
            raise HypatiaHipError("no HIP device visible: the hypatia.jl_amd path needs an MI355X (no CPU fallback)")
        dev = int(os.environ.get("LOCAL_RANK", "0")) % n.value
        h = c_vp()
        rc = L.hyp_ctx_create(dev, ctypes.byref(h))
        if rc != 0:
            raise HypatiaHipError("hyp_ctx_create failed (%d): %s" % (rc, L.hyp_last_error(None).decode()))
        _ctx = h
    return _ctx


def last_error():
    """text of the calling thread's last library error ("" if none)"""
    return _lib.hyp_last_error(_ctx).decode() if _lib is not None else ""


def check(rc, what=""):
    if rc != 0:
        msg = _lib.hyp_last_error(_ctx).decode() if _lib is not None else ""
        raise HypatiaHipError("%s failed (%d): %s" % (what, rc, msg))


def vec_ptr(a):
    """pointer to a contiguous float64 vector (must already be contiguous: written in place)."""
    assert a.dtype == np.float64 and a.ndim == 1 and (a.shape[0] <= 1 or a.strides[0] == 8), "need a contiguous float64 vector"
    return a.ctypes.data_as(c_vp)


def mat_view(a):
    """(array, pointer, ld, ncols, is_copy) for a (dim x ncols) column-major operand."""
    if a.ndim == 1:
        assert a.dtype == np.float64 and (a.shape[0] <= 1 or a.strides[0] == 8)
        return a, a.ctypes.data_as(c_vp), a.shape[0], 1, False
    assert a.ndim == 2 and a.dtype == np.float64
    dim, nc = a.shape
    ok = (dim <= 1 or a.strides[0] == 8) and (nc <= 1 or (a.strides[1] % 8 == 0 and a.strides[1] >= 8 * dim))
    if ok:
        ld = a.strides[1] // 8 if nc > 1 else max(dim, 1)
        return a, a.ctypes.data_as(c_vp), ld, nc, False
    b = np.asfortranarray(a)
    return b, b.ctypes.data_as(c_vp), max(dim, 1), nc, True
