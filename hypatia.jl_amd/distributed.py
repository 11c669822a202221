"""Multi-GPU path: cone-sharded interior-point iterations, one process per GPU.

SURVEY.md 8(e): `lhs = sum_k G_k' (mu H_k) G_k` is a sum over cones (qrchol.jl:219-246) and cone oracles
are independent per cone (search.jl:118-134), so cones -- and with them the matching row blocks of G,
h, z, s -- are partitioned across ranks.  Every rank runs the same (replicated, deterministic) driver on
the full point vector; cone k's oracles run on its owner and the result is broadcast; products with G
are local row-block products followed by one all-reduce; the only large exchange is the all-reduce
(sum, f64) of the n x n Schur matrix once per iteration, after which every rank factors it.

Collectives go through `torch.distributed`: backend "nccl" (= RCCL over xGMI) on the GPU box, "gloo" in
the CPU tests.  The local numerical back end is injected (`local_backend`): the HIP system solver /
cones here, the numpy oracle in tests/test_distributed_gloo.py (there is no CPU fallback in the
product: the default back end needs the GPU).
"""
import sys

import numpy as np

from . import _lib as L
from .models import Model
from .systemsolvers import QRCholDenseSystemSolver, SubPoint

EPS = np.finfo(np.float64).eps


class Comm:
    """numpy-facing wrapper over an initialised torch.distributed process group."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        assert dist.is_initialized(), "call torch.distributed.init_process_group first"
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device or ("cuda" if dist.get_backend() == "nccl" else "cpu")
        self._ops = {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}
        self.n_collectives = 0
        self.payload_log = []
        self.hist = None     # HYP_PROFILE=1: {(origin, payload doubles, op): calls}
        import os
        if os.environ.get("HYP_PROFILE", "0") not in ("", "0"):
            import collections
            self.hist = collections.Counter()

    def _count(self, origin, count, op):
        self.n_collectives += 1
        if str(origin).startswith("host"):      # (the library's own exchanges are counted by hyp_sys_comm_stats)
            self.payload_log.append(int(count))   # doubles per host-level collective, in order: tests assert what an iteration exchanges
        if self.hist is not None:
            self.hist[(origin, int(count), str(op))] += 1

    def _to(self, arr):
        t = self.torch.from_numpy(np.ascontiguousarray(arr))
        return t.to(self.device) if self.device != "cpu" else t.clone()

    def allreduce(self, arr, op="sum"):
        """in-place all-reduce of a float64 numpy array"""
        t = self._to(arr)
        self.dist.all_reduce(t, op=self._ops[op])
        arr[...] = t.cpu().numpy().reshape(arr.shape)
        self._count("host:" + (sys._getframe(1).f_code.co_name if self.hist is not None else ""), arr.size, op)
        return arr

    def bcast(self, arr, src):
        t = self._to(arr)
        self.dist.broadcast(t, src=src)
        arr[...] = t.cpu().numpy().reshape(arr.shape)
        self._count("host:bcast", arr.size, "bcast")
        return arr

    def bcast_scalar(self, value, src):
        a = np.array([float(value)])
        return float(self.bcast(a, src)[0])

    def barrier(self):
        self.dist.barrier()


def init_library_rccl(comm, sys_handle, lib=None, lib_ctx=None):
    """RCCL inside the library for one hyp_sys: rank 0 creates the unique id (hyp_comm_unique_id), torch.distributed only carries
    its 128 bytes, every rank joins (hyp_comm_init_rank) and hands the communicator to its solver (hyp_sys_set_comm_rccl).
    Every step is AGREED on by all ranks (a MIN all-reduce of the local outcome) before the next collective is entered, so a
    failure on one rank -- the id, ncclCommInitRank, a self-check all-reduce whose sum is known -- can neither raise on that rank
    alone nor leave the others waiting: all ranks return None together and the caller falls back to the callback transport.
    Returns the hyp_comm handle (to be released with release_library_rccl) or None.
    lib / lib_ctx: the library binding and its context; the world-2 CPU test of THIS protocol (tests/test_distributed_gloo.py)
    injects a stand-in that fails where it is told to, over gloo -- the product passes nothing and needs the nccl back end."""
    import ctypes
    import os
    torch, dist = comm.torch, comm.dist
    injected = lib is not None
    if not injected and (dist.get_backend() != "nccl" or os.environ.get("HYP_DIST_RCCL", "1") in ("0",)):
        return None
    if not injected:
        lib, lib_ctx = L.lib(), L.ctx()
    dev = "cuda" if comm.device == "cuda" else "cpu"
    sync = torch.cuda.synchronize if dev == "cuda" else (lambda: None)

    def all_ok(ok):
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    uid = ctypes.create_string_buffer(128)
    ok = True
    if comm.rank == 0:
        ok = lib.hyp_comm_unique_id(uid) == 0
    box = [uid.raw]
    dist.broadcast_object_list(box, src=0)
    why = "hyp_comm_unique_id"
    hc = ctypes.c_void_p()
    if all_ok(ok):
        uid = ctypes.create_string_buffer(box[0], 128)
        ok = lib.hyp_comm_init_rank(lib_ctx, comm.world, comm.rank, uid, ctypes.byref(hc)) == 0
        why = "hyp_comm_init_rank"
        if all_ok(ok):
            # start-up self-check: sum of (rank + 1) over the new communicator, on a device buffer, before anything depends on it
            t = torch.full((8,), float(comm.rank + 1), dtype=torch.float64, device=dev)
            sync()       # (the fill ran on torch's stream, the all-reduce runs on the library's)
            ok = lib.hyp_comm_allreduce(hc, ctypes.c_void_p(t.data_ptr()), 8, 0) == 0
            sync()
            ok = ok and bool(torch.all(t == comm.world * (comm.world + 1) / 2.0).item())
            why = "self-check all-reduce"
            if all_ok(ok):
                rc = lib.hyp_sys_set_comm_rccl(sys_handle, hc)
                if not injected:
                    L.check(rc, "hyp_sys_set_comm_rccl")
                return hc
        if hc.value:
            lib.hyp_comm_destroy(hc)
    if comm.rank == 0:
        print("hypatia_jl_amd: RCCL inside the library is not available (%s failed on some rank); collectives go through "
              "torch.distributed" % why)
    return None


def release_library_rccl(sys_handle, hc):
    """detach the communicator from the solver, then destroy it (ncclCommDestroy)"""
    if hc is None:
        return
    try:
        lib = L.lib()
        if sys_handle is not None:
            lib.hyp_sys_set_comm_rccl(sys_handle, None)
        lib.hyp_comm_destroy(hc)
    except Exception:
        pass


def partition_cones(ncones, world):
    """contiguous blocks of cones per rank (config 4: 64 cones -> 8 per GPU)"""
    base, rem = divmod(ncones, world)
    owners = []
    for r in range(world):
        owners += [r] * (base + (1 if r < rem else 0))
    return owners


class ShardedCone:
    """Proxy with the Cone interface (Cones.jl:27-310): the owner rank runs the oracle on its local cone
    (a HIP cone on the GPU), every rank receives the result.  All ranks must call in the same order."""

    def __init__(self, comm, owner, local_cone, dim, nu, use_dual_barrier=False, is_nonnegative=False):
        self.comm, self.owner, self.local = comm, owner, local_cone
        self.mine = (comm.rank == owner)
        assert self.mine == (local_cone is not None)
        self.dim, self.nu = dim, nu
        self._udb = use_dual_barrier
        self.is_nonnegative = is_nonnegative
        self.setup_data()

    def dimension(self): return self.dim
    def get_nu(self): return self.nu
    def use_dual_barrier(self): return self._udb
    def use_dder3(self): return True

    def setup_data(self):
        d = self.dim
        self.point, self.dual_point = np.zeros(d), np.zeros(d)
        self.grad, self.dder3_ = np.zeros(d), np.zeros(d)
        self.vec1, self.vec2 = np.zeros(d), np.zeros(d)
        self._grad_valid = False
        if self.mine:
            self.local.setup_data()
        return self

    def set_initial_point(self, arr):
        tmp = np.zeros(self.dim)
        if self.mine:
            self.local.set_initial_point(tmp)
        arr[:] = self.comm.bcast(tmp, self.owner)
        return arr

    def load_point(self, point, scal=None):
        if scal is None:
            self.point[:] = point
        else:
            np.multiply(point, scal, out=self.point)
        if self.mine:
            self.local.load_point(np.ascontiguousarray(point), scal)

    def load_dual_point(self, point):
        self.dual_point[:] = point
        if self.mine:
            self.local.load_dual_point(np.ascontiguousarray(point))

    def reset_data(self):
        self._grad_valid = False
        if self.mine:
            self.local.reset_data()

    def _flag(self, name, *a):
        v = float(getattr(self.local, name)(*a)) if self.mine else 0.0
        return bool(self.comm.bcast_scalar(v, self.owner))

    def is_feas(self): return self._flag("is_feas")
    def is_dual_feas(self): return self._flag("is_dual_feas")
    def check_numerics(self): return self._flag("check_numerics")

    def use_sqrt_hess_oracles(self, arr_dim):
        return self._flag("use_sqrt_hess_oracles", arr_dim)

    def get_grad(self):
        if not self._grad_valid:
            if self.mine:
                self.grad[:] = self.local.get_grad()
            self.comm.bcast(self.grad, self.owner)
            self._grad_valid = True
        return self.grad

    def _prod(self, name, prod, arr):
        tmp = np.zeros(np.shape(arr), order="F")
        if self.mine:
            getattr(self.local, name)(tmp, np.asfortranarray(arr) if np.ndim(arr) == 2 else np.ascontiguousarray(arr))
        self.comm.bcast(tmp, self.owner)
        prod[...] = tmp
        return prod

    def hess_prod(self, prod, arr): return self._prod("hess_prod", prod, arr)
    def inv_hess_prod(self, prod, arr): return self._prod("inv_hess_prod", prod, arr)
    def hess_prod_slow(self, prod, arr): return self._prod("hess_prod_slow", prod, arr)
    def sqrt_hess_prod(self, prod, arr): return self._prod("sqrt_hess_prod", prod, arr)
    def inv_sqrt_hess_prod(self, prod, arr): return self._prod("inv_sqrt_hess_prod", prod, arr)

    def dder3(self, dir):
        if self.mine:
            self.dder3_[:] = self.local.dder3(np.ascontiguousarray(dir))
        self.comm.bcast(self.dder3_, self.owner)
        return self.dder3_

    def update_hess_aux(self):
        if self.mine:
            self.local.update_hess_aux()

    def get_proxsqr(self, irtmu, use_max_prox):
        v = self.local.get_proxsqr(irtmu, use_max_prox) if self.mine else 0.0
        return self.comm.bcast_scalar(v, self.owner)


class DistModel:
    """Models.Model with G held as this rank's row blocks only (rows of the cones it owns)."""

    def __init__(self, comm, c, h, G_local, cones, owners, obj_offset=0.0):
        self.comm = comm
        self.c = np.array(c, dtype=np.float64)
        self.h = np.array(h, dtype=np.float64)
        self.b = np.zeros(0)
        self.n, self.p, self.q = self.c.shape[0], 0, self.h.shape[0]
        self.A = np.zeros((0, self.n))
        self.obj_offset = float(obj_offset)
        self.cones, self.owners = list(cones), list(owners)
        self.cone_idxs = []
        prev = 0
        for cone in self.cones:
            self.cone_idxs.append(slice(prev, prev + cone.dimension()))
            prev += cone.dimension()
        assert prev == self.q
        self.nu = float(sum(cone.get_nu() for cone in self.cones))
        self.local_ks = [k for k, o in enumerate(self.owners) if o == comm.rank]
        self.local_rows = (np.concatenate([np.arange(self.cone_idxs[k].start, self.cone_idxs[k].stop) for k in self.local_ks])
                           if self.local_ks else np.zeros(0, dtype=int))
        self.G_local = np.asfortranarray(G_local, dtype=np.float64).reshape(self.local_rows.shape[0], self.n)
        self.G = None   # never materialised: products go through the system solver
        self.dist_hooks = DistHooks(comm, self)

    def copy(self):
        return DistModel(self.comm, self.c, self.h, self.G_local.copy(order="F"), self.cones, self.owners, self.obj_offset)


# ---------------------------------------------------------------------------------------------------
# local numerical back end on the GPU (one hyp_sys over this rank's cones and rows)
# ---------------------------------------------------------------------------------------------------
class HipLocalSys:
    def __init__(self, comm, model):
        import ctypes
        self.ct = ctypes
        self.comm = comm
        self.n = model.n
        self.q_local = model.local_rows.shape[0]
        lib = L.lib()
        locs = [model.cones[k].local for k in model.local_ks]
        handles = (ctypes.c_void_p * max(len(locs), 1))(*[c._h for c in locs])
        h = ctypes.c_void_p()
        L.check(lib.hyp_sys_create(L.ctx(), self.n, 0, self.q_local, handles, len(locs), ctypes.byref(h)), "hyp_sys_create")
        self._h = h
        G = np.asfortranarray(model.G_local)
        L.check(lib.hyp_sys_load(h, G.ctypes.data_as(ctypes.c_void_p), None, None, None, None), "hyp_sys_load")
        self._lhs_dev = None
        if comm.device == "cuda":
            self._lhs_dev = comm.torch.empty(self.n * self.n, dtype=comm.torch.float64, device="cuda")

    def __del__(self):
        try:
            if self._h is not None and L._lib is not None:
                L._lib.hyp_sys_destroy(self._h)
        except Exception:
            pass

    def assemble_lhs(self):
        L.check(L.lib().hyp_sys_assemble_lhs(self._h, None), "hyp_sys_assemble_lhs")

    def allreduce_lhs(self):
        lib, ct = L.lib(), self.ct
        if self._lhs_dev is not None:   # RCCL on the device buffer, no host round trip
            L.check(lib.hyp_sys_lhs_export_dev(self._h, ct.c_void_p(self._lhs_dev.data_ptr())), "lhs_export")
            self.comm.dist.all_reduce(self._lhs_dev)
            self.comm.torch.cuda.synchronize()
            L.check(lib.hyp_sys_lhs_import_dev(self._h, ct.c_void_p(self._lhs_dev.data_ptr())), "lhs_import")
            self.comm.n_collectives += 1
        else:
            buf = np.zeros((self.n, self.n), order="F")
            L.check(lib.hyp_sys_get_lhs(self._h, buf.ctypes.data_as(ct.c_void_p)), "get_lhs")
            self.comm.allreduce(buf)
            L.check(lib.hyp_sys_set_lhs(self._h, buf.ctypes.data_as(ct.c_void_p)), "set_lhs")

    def factor_lhs(self):
        ct = self.ct
        info, fb = ct.c_int(0), ct.c_int(0)
        L.check(L.lib().hyp_sys_factor_lhs(self._h, ct.byref(info), ct.byref(fb)), "hyp_sys_factor_lhs")
        return info.value, bool(fb.value)

    def potrs(self, x):
        L.check(L.lib().hyp_sys_potrs(self._h, L.vec_ptr(x)), "hyp_sys_potrs")
        return x

    def mul_G(self, trans, x, out):
        """out = G_local x (trans False: x is n, out q_local) or G_local' x (trans True)"""
        if self.q_local == 0:
            out[:] = 0
            return out
        L.check(L.lib().hyp_sys_mul_G(self._h, int(trans), 1.0, L.vec_ptr(np.ascontiguousarray(x)), 0.0, L.vec_ptr(out)), "hyp_sys_mul_G")
        return out

    def block_hess_prod(self, out_local, in_local):
        if self.q_local:
            L.check(L.lib().hyp_sys_block_hess_prod(self._h, L.vec_ptr(out_local), L.vec_ptr(np.ascontiguousarray(in_local))), "block_hess_prod")
        return out_local

    def get_lhs(self):
        out = np.zeros((self.n, self.n), order="F")
        L.check(L.lib().hyp_sys_get_lhs(self._h, out.ctypes.data_as(self.ct.c_void_p)), "hyp_sys_get_lhs")
        return out


class DistQRCholDenseSystemSolver(QRCholDenseSystemSolver):
    """QRCholDenseSystemSolver over cone-sharded data (p = 0, i.e. the default reduce = true path)."""

    one_pass_residual_products = False   # (its residual_products is the row-local form; calc_convergence_params takes that branch itself)

    def __init__(self, comm, local_backend=HipLocalSys):
        super().__init__()
        self.comm = comm
        self.local_backend = local_backend

    def load(self, solver):
        self.__dict__.pop("_screen_usable", None)   # (the candidate screen is decided per loaded model)
        self._dirs_resident = False
        model = solver.model
        assert model.p == 0, "the sharded solver assumes the reduced model (p = 0)"
        self.n, self.p, self.q = model.n, 0, model.q
        self.model = model
        self.local = self.local_backend(self.comm, model)
        self.rows = model.local_rows
        self.use_sqrt_hess_cones = [True] * len(model.cones)
        self.sol_sub, self.rhs_sub = SubPoint(model), SubPoint(model)
        self.rhs_const, self.sol_const = SubPoint(model), SubPoint(model)
        self.rhs_const.x[:] = -model.c
        self.rhs_const.z[:] = model.h
        self.last_info, self.used_fallback = 0, False
        self._ql = np.zeros(self.rows.shape[0])
        self._ql2 = np.zeros(self.rows.shape[0])
        self._install_native(model)
        return self

    # ---- device-resident distributed step: the rank's hyp_sys runs the fused native routines on ITS rows and
    # cones; at the exchange points it calls back here and the payload (already in a torch device tensor) is
    # all-reduced in place over RCCL.  HYP_DIST_NATIVE=0 keeps the host-composed path.
    native_directions = False
    native_caps = frozenset()

    def _install_native(self, model):
        import os
        self._hooked = False
        if not isinstance(self.local, HipLocalSys) or self.comm.device != "cuda" or os.environ.get("HYP_DIST_NATIVE", "1") in ("0",):
            return
        import ctypes
        torch, dist = self.comm.torch, self.comm.dist
        n = self.n
        self._stage = torch.empty(max(n * n, 1024), dtype=torch.float64, device="cuda")
        ops = {0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MAX, 2: dist.ReduceOp.MIN}
        comm, stage = self.comm, self._stage

        def _allreduce(user, count, op):
            try:
                dist.all_reduce(stage[:count], op=ops[op])
                torch.cuda.synchronize()
                comm._count("library", count, op)
                return 0
            except Exception as e:   # never let an exception cross the C boundary
                print("all-reduce callback failed:", e)
                return 1

        self._cb = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_int)(_allreduce)
        lib, h = L.lib(), self.local._h
        self.rccl_in_library = False
        self._hyp_comm = init_library_rccl(self.comm, h)   # (agreed on by all ranks; None: the callback below)
        self.rccl_in_library = self._hyp_comm is not None
        if not self.rccl_in_library:
            L.check(lib.hyp_sys_set_comm(h, ctypes.cast(self._cb, ctypes.c_void_p), None, ctypes.c_void_p(stage.data_ptr()), int(stage.numel())),
                    "hyp_sys_set_comm")
        # (rank / world of the transport: lets the library send the scalars of a solve behind its n-vectors -- one exchange for three;
        #  the RCCL communicator carries its own layout, the callback does not)
        L.check(lib.hyp_sys_set_comm_layout(h, int(self.comm.rank), int(self.comm.world)), "hyp_sys_set_comm_layout")
        cc = np.ascontiguousarray(model.c, dtype=np.float64)
        hl = np.ascontiguousarray(model.h[self.rows], dtype=np.float64)
        bb = np.zeros(1)
        L.check(lib.hyp_sys_load_model(h, L.vec_ptr(cc), L.vec_ptr(bb), L.vec_ptr(hl), None), "hyp_sys_load_model")
        self._hooked = True
        self.native_directions = True
        self.native_caps = frozenset({"fused", "search"})
        self.cand_in_temp = False
        self._ql_n = self.rows.shape[0]
        # row-local driver state (solvers.py: calc_mu, calc_convergence_params, the accepted candidate): on this rank only ITS
        # rows of the z / s vectors are maintained -- no q-vector is exchanged inside an iteration.  Needs contiguous rows
        # (partition_cones deals out contiguous blocks of cones).
        r = self.rows
        contiguous = r.shape[0] > 0 and np.array_equal(r, np.arange(r[0], r[0] + r.shape[0]))
        # The row-local state is only read by the FUSED step (solvers.py CombinedStepper.step): the unfused branches
        # (HYP_NO_FUSED_STEP=1 / HYP_NO_PAIR=1) build their right-hand sides from full-length point.z / point.s / z_residual,
        # so with either switch the driver stays replicated.
        unfused = any(os.environ.get(k, "0") not in ("", "0") for k in ("HYP_NO_FUSED_STEP", "HYP_NO_PAIR"))
        if contiguous and not unfused and os.environ.get("HYP_DIST_ROW_LOCAL", "1") not in ("0",):
            self.rsl = slice(int(r[0]), int(r[0]) + int(r.shape[0]))
            self.row_local = True
            self._last_cand = None

    row_local = False

    def agree_any(self, flag):
        """True on every rank iff `flag` is true on some rank (decisions that depend on a rank's own clock)"""
        v = np.array([1.0 if flag else 0.0])
        self.comm.allreduce(v, "max")
        return bool(v[0] > 0.5)

    def close(self):
        """release the library's communicator (every rank, before the process group goes away)"""
        loc = getattr(self, "local", None)
        release_library_rccl(getattr(loc, "_h", None), getattr(self, "_hyp_comm", None))
        self._hyp_comm = None

    def point_version(self, pt):
        """fingerprint of this rank's part of a point: the record calc_mu leaves for calc_convergence_params is only reused for
        the very point it was computed at"""
        z, sv = pt.z[self.rsl], pt.s[self.rsl]
        return (id(pt), float(pt.tau), float(pt.kap), float(pt.x @ pt.x), float(z @ z), float(sv @ sv))

    def residual_products(self, pt):
        """hyp_sys_residual_products2 on this rank's rows: {Gtz (n, summed over ranks), Gx_s (local rows), hz, zs (summed), and the two
        residual norms of Solvers.jl:447-457 over all ranks' rows (zn_t = max |G x + s|, zn = max |G x + s - h tau|)} -- ONE exchange"""
        import ctypes
        n, ql = self.n, self._ql_n
        Gtz, Gx_s, dots, norms = np.zeros(n), np.zeros(max(ql, 1)), np.zeros(2), np.zeros(2)
        z = np.ascontiguousarray(pt.z[self.rsl])
        sv = np.ascontiguousarray(pt.s[self.rsl])
        L.check(L.lib().hyp_sys_residual_products2(self.local._h, L.vec_ptr(np.ascontiguousarray(pt.x)), L.vec_ptr(z), L.vec_ptr(sv),
                                                   ctypes.c_double(float(pt.tau)), L.vec_ptr(Gtz), L.vec_ptr(Gx_s), L.vec_ptr(dots),
                                                   L.vec_ptr(norms)), "hyp_sys_residual_products2")
        return {"Gtz": Gtz, "Gx_s": Gx_s[:ql], "hz": float(dots[0]), "zs": float(dots[1]), "zn_t": float(norms[0]), "zn": float(norms[1]),
                "tau": float(pt.tau), "version": self.point_version(pt)}

    def comm_hist(self):
        """exchanges the library has issued for this solver, by place in the iteration (hyp_sys_comm_hist)"""
        import ctypes
        out = (ctypes.c_longlong * 16)()
        L.check(L.lib().hyp_sys_comm_hist(self.local._h, out), "hyp_sys_comm_hist")
        return [int(v) for v in out]

    def reduce_max(self, vals):
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        L.check(L.lib().hyp_sys_allreduce_host(self.local._h, L.vec_ptr(vals), int(vals.shape[0]), 1), "hyp_sys_allreduce_host")
        return vals

    def accept_candidate(self, pt):
        cand, ql = self._last_cand, self._ql_n
        assert cand is not None
        pt.z[self.rsl] = cand[:ql]
        pt.tau = cand[ql]
        pt.s[self.rsl] = cand[ql + 1:2 * ql + 1]
        pt.kap = cand[2 * ql + 1]

    def gather_rows(self, pt):
        for v in (pt.z, pt.s):
            full = np.zeros(self.q)
            full[self.rsl] = v[self.rsl]
            self.comm.allreduce(full)
            v[:] = full

    def _local_ztsk(self, pt):
        rows = self.rsl if self.row_local else self.rows
        return np.concatenate([pt.z[rows], [pt.tau], pt.s[rows], [pt.kap]])

    def step_directions_native(self, solver, stepper):
        import ctypes
        model, n, ql = solver.model, self.n, self._ql_n
        rows = self.rsl if self.row_local else self.rows     # (a slice: views, no gather of 2 q_local entries per vector)
        c_int = ctypes.c_int
        pt = solver.point
        vec_l = np.concatenate([pt.x, pt.z[rows], [pt.tau], pt.s[rows], [pt.kap]])
        res_l = np.concatenate([solver.x_residual, solver.z_residual[rows]])
        dv_l = n + 2 * ql + 2
        if getattr(self, "_dirs_l", None) is None or self._dirs_l.shape != (4, dv_l):
            self._dirs_l = np.zeros((4, dv_l))      # (kept: 13 MB of fresh zeros per iteration at q_local = 207 360 otherwise)
        dirs_l = self._dirs_l
        if not hasattr(self, "_dir_rows_set"):
            # row-local driver with the resident line search: of the four directions the host reads the x rows and tau / kap only
            # (update_stepper_points_x; the accepted candidate's z / tau / s / kap rows come back from the search): only those are
            # downloaded (hyp_sys_set_direction_rows).  Decided once per load, the same on every rank (_screen_ok is agreed).
            import os
            self._x_only = bool(self.row_local and os.environ.get("HYP_DIRS_X_ONLY", "1") != "0" and self._screen_ok())
            L.check(L.lib().hyp_sys_set_direction_rows(self.local._h, 1 if self._x_only else 0), "hyp_sys_set_direction_rows")
            self._dir_rows_set = True
        nc_l = len(model.local_ks)
        flags = (c_int * max(nc_l, 1))()
        info, fb, ns = c_int(0), c_int(0), c_int(0)
        resn = (ctypes.c_double * 4)()
        L.check(L.lib().hyp_sys_step_directions(self.local._h, L.vec_ptr(vec_l), L.vec_ptr(res_l), float(solver.tau_residual), float(solver.mu),
                                                int(solver.max_ref_steps), float(solver.res_norm_cutoff), 0.5,
                                                dirs_l.ctypes.data_as(ctypes.c_void_p), resn, ctypes.byref(ns), flags, ctypes.byref(info),
                                                ctypes.byref(fb), None), "hyp_sys_step_directions")
        self.last_info, self.used_fallback = info.value, bool(fb.value)
        self.fallback_kind = fb.value   # 0 Cholesky, 1 Bunch-Kaufman, 2 diagonal shift + Bunch-Kaufman
        if info.value != 0:
            print("positive definite linear system factorization failed")
            return False
        q = self.q
        if self.row_local:   # every rank keeps its own rows of the directions; nobody needs the others'
            for k, d in enumerate((stepper.dir_cent, stepper.dir_pred, stepper.dir_centadj, stepper.dir_predadj)):
                d.x[:] = dirs_l[k, :n]
                if not self._x_only:
                    d.z[self.rsl] = dirs_l[k, n:n + ql]
                    d.s[self.rsl] = dirs_l[k, n + ql + 1:n + 2 * ql + 1]
                d.tau = dirs_l[k, n + ql]
                d.kap = dirs_l[k, -1]
        else:
            # global z / s parts of the four directions: every rank contributes its rows, one all-reduce
            zs = np.zeros((4, 2 * q))
            for k in range(4):
                zs[k, rows] = dirs_l[k, n:n + ql]
                zs[k, q + rows] = dirs_l[k, n + ql + 1:n + 2 * ql + 1]
            self.comm.allreduce(zs)
            for k, d in enumerate((stepper.dir_cent, stepper.dir_pred, stepper.dir_centadj, stepper.dir_predadj)):
                d.x[:] = dirs_l[k, :n]
                d.z[:] = zs[k, :q]
                d.s[:] = zs[k, q:]
                d.tau = dirs_l[k, n + ql]
                d.kap = dirs_l[k, -1]
        solver.n_solves += ns.value
        assert not any(np.isnan(resn[k]) for k in range(4))
        if solver.max_ref_steps > 0:
            solver.worst_dir_res = max(solver.worst_dir_res, *[resn[k] for k in range(4)])
        self._dirs_resident = True   # (this rank's rows of the point and the directions stay on the device: search_alpha_native)
        return True

    def _screen_ok(self):
        """the side-by-side candidate screen applies on EVERY rank (each rank's cones one run of equal PosSemidefTri cones): then
        the walk may form its candidates from the device-resident rows (hyp_sys_search_alpha_resident); agreed once per load"""
        if not hasattr(self, "_screen_usable"):
            import ctypes, os
            u, a, b = ctypes.c_int(0), ctypes.c_longlong(0), ctypes.c_longlong(0)
            L.check(L.lib().hyp_sys_search_screen_stats(self.local._h, ctypes.byref(u), ctypes.byref(a), ctypes.byref(b)),
                    "hyp_sys_search_screen_stats")
            ok = bool(u.value) and os.environ.get("HYP_SEARCH_RESIDENT", "1") != "0"
            self._screen_usable = self.reduce_max([0.0 if ok else 1.0])[0] < 0.5
        return self._screen_usable

    def last_update_lhs_seconds(self):
        import ctypes
        out = ctypes.c_double(0.0)
        L.check(L.lib().hyp_sys_last_update_lhs_seconds(self.local._h, ctypes.byref(out)), "hyp_sys_last_update_lhs_seconds")
        return out.value

    def search_alpha_native(self, model, point, stepper, sched):
        import ctypes
        c_int = ctypes.c_int
        searcher = stepper.searcher
        sc = np.ascontiguousarray(searcher.alpha_sched, dtype=np.float64)
        ql = self._ql_n
        cand = np.zeros(2 * ql + 2)
        idx, nt, nl = c_int(-1), c_int(0), c_int(0)
        prox, irtmu = ctypes.c_double(0.0), ctypes.c_double(0.0)
        if getattr(self, "_dirs_resident", False) and self._screen_ok():
            # this rank's rows of the point and of the directions are still on the device (step_directions_native): the schedule's
            # candidates are formed and screened there; two small all-reduces make the verdicts the same on every rank
            L.check(L.lib().hyp_sys_search_alpha_resident(
                self.local._h, int(stepper.unadj_only), int(stepper.cent_only), L.vec_ptr(sc), len(sc), int(sched - 1),
                float(searcher.min_prox), float(searcher.prox_bound), int(bool(searcher.use_max_prox)), float(searcher.nup1),
                L.vec_ptr(cand), ctypes.byref(idx), ctypes.byref(prox), ctypes.byref(nt), ctypes.byref(nl), ctypes.byref(irtmu)),
                "hyp_sys_search_alpha_resident")
        else:
            loc = [self._local_ztsk(p) for p in (point, stepper.dir_cent, stepper.dir_pred, stepper.dir_centadj, stepper.dir_predadj)]
            L.check(L.lib().hyp_sys_search_alpha(
                self.local._h, L.vec_ptr(loc[0]), L.vec_ptr(loc[1]), L.vec_ptr(loc[2]), L.vec_ptr(loc[3]), L.vec_ptr(loc[4]),
                int(stepper.unadj_only), int(stepper.cent_only), L.vec_ptr(sc), len(sc), int(sched - 1), float(searcher.min_prox),
                float(searcher.prox_bound), int(bool(searcher.use_max_prox)), float(searcher.nup1), L.vec_ptr(cand), ctypes.byref(idx),
                ctypes.byref(prox), ctypes.byref(nt), ctypes.byref(nl), ctypes.byref(irtmu)), "hyp_sys_search_alpha")
        searcher.n_trials += nt.value
        self._last_cand = cand
        # host mirrors of this rank's cones follow the last candidate they were loaded with
        off = 0
        for j, k in enumerate(model.local_ks):
            cone = model.cones[k].local
            dk = cone.dim
            if j < nl.value:
                zk, sk = cand[off:off + dk], cand[ql + 1 + off:ql + 1 + off + dk]
                prim, dual = (zk, sk) if cone.use_dual_barrier() else (sk, zk)
                cone._mirror_loaded(prim, irtmu.value, dual)
            off += dk
        if idx.value >= 0:
            searcher.prox = prox.value
            searcher.prev_sched = idx.value + 1
            return float(sc[idx.value])
        searcher.prev_sched = len(sc) + 1
        return 0.0

    # y = alpha op(G) x + beta y over ALL rows: local row-block product + one all-reduce
    def mul_G(self, trans, x, alpha=1.0, beta=0.0, y=None):
        if trans:
            part = np.zeros(self.n)
            self.local.mul_G(True, np.ascontiguousarray(np.asarray(x)[self.rows]), part)
            self.comm.allreduce(part)
        else:
            part = np.zeros(self.q)
            self.local.mul_G(False, x, self._ql)
            part[self.rows] = self._ql
            self.comm.allreduce(part)
        if y is None:
            return alpha * part
        y[:] = alpha * part + (beta * y if beta != 0.0 else 0.0)
        return y

    def block_hess_prod_full(self, out_full, in_full):
        """block_hess_prod!.(out_k, in_k, cones) on a full q-vector: owners compute, one all-reduce"""
        out_full[:] = 0
        self.local.block_hess_prod(self._ql2, np.ascontiguousarray(in_full[self.rows]))
        out_full[self.rows] = self._ql2
        self.comm.allreduce(out_full)
        return out_full

    def update_lhs_fact(self, solver):
        self.local.assemble_lhs()        # sum over this rank's cones (with the native hook installed: already all-reduced)
        if not getattr(self, "_hooked", False):
            self.local.allreduce_lhs()   # the one large exchange: n x n, sum, f64
        self.last_info, self.used_fallback = self.local.factor_lhs()
        if self.last_info != 0:
            print("positive definite linear system factorization failed")

    def update_lhs(self, solver):
        model = solver.model
        self.update_lhs_fact(solver)
        if self.last_info != 0:   # (no factorization: the stepper ends in NumericalFailure on every rank -- the same matrix everywhere)
            return self
        self.block_hess_prod_full(self.rhs_const.z, model.h)
        self.solve_subsystem3(solver, self.sol_const, self.rhs_const)
        return self

    def setup_rhs3(self, model, rhs, sol, rhs_sub):   # qrchol.jl:16-37 (primal-barrier cones)
        tmp = np.zeros(self.q)
        self.block_hess_prod_full(tmp, rhs.z)
        rhs_sub.z[:] = -rhs.s - tmp

    def solve_subsystem3(self, solver, sol, rhs):   # qrchol.jl:39-85 with p = 0, Ap_Q = I
        sol.vec[:] = rhs.vec
        x, z = sol.x, sol.z
        t = self.mul_G(True, z)
        t += x
        self.local.potrs(t)
        x[:] = t
        Gx = self.mul_G(False, x)
        HGx = np.zeros(self.q)
        self.block_hess_prod_full(HGx, Gx)
        z[:] = HGx - z
        return sol


# ---------------------------------------------------------------------------------------------------
# per-cone loops of the driver, batched: every rank evaluates the oracles of ITS cones concurrently and one
# all-reduce assembles the result (the proxies above would run the cones one after another, each with its
# own broadcast: the line search alone is 4 collectives per cone per trial)
# ---------------------------------------------------------------------------------------------------
class DistHooks:
    def __init__(self, comm, model):
        self.comm, self.model = comm, model

    def _locals(self):
        m = self.model
        return [(k, m.cones[k].local) for k in m.local_ks]

    # steppers/common.jl:63-84
    def update_rhs_cent(self, solver, rhs):
        rhs.x[:] = 0; rhs.y[:] = 0; rhs.z[:] = 0; rhs.tau = 0
        rtmu = np.sqrt(solver.mu)
        rhs.s[:] = 0
        for k, cone in self._locals():
            rhs.s_views[k][:] = -solver.point.dual_views[k] - rtmu * cone.get_grad()
        self.comm.allreduce(rhs.s)
        rhs.kap = -solver.point.kap + solver.mu / solver.point.tau
        return rhs

    # steppers/common.jl:27-60 (adj = "pred") and :87-118 (adj = "cent")
    def update_rhs_adj(self, solver, rhs, dir, which):
        rhs.vec[:] = 0
        rteps = np.sqrt(np.finfo(np.float64).eps)
        irtrtmu = 1.0 / np.sqrt(np.sqrt(solver.mu))
        for k, cone in self._locals():
            prim_dir_k = dir.primal_views[k]
            scal = irtrtmu * prim_dir_k
            H = np.zeros(cone.dim)
            if which == "pred":
                cone.hess_prod_slow(H, np.ascontiguousarray(prim_dir_k))
                d3 = cone.dder3(scal)
                dot1 = d3 @ cone.point
                dot2 = irtrtmu * (scal @ H)
                if abs(dot1 - dot2) / (rteps + abs(dot2)) < 1e-4:
                    rhs.s_views[k][:] = H + d3
            else:
                cone.hess_prod_slow(H, scal)
                d3 = cone.dder3(scal)
                dot1 = d3 @ cone.point
                dot2 = scal @ H
                if abs(dot1 - dot2) / (rteps + abs(dot2)) < 1e-4:
                    rhs.s_views[k][:] = d3
        self.comm.allreduce(rhs.s)
        taubar = solver.point.tau
        t = dir.tau / taubar
        rhs.kap = t * solver.mu / taubar * ((1 + t) if which == "pred" else t)
        return rhs

    # the cone rows of apply_lhs (systemsolvers/common.jl:107-115)
    def apply_lhs_cones(self, res, dir):
        res.s[:] = 0
        for k, cone in self._locals():
            out = np.zeros(cone.dim)
            cone.hess_prod_slow(out, np.ascontiguousarray(dir.primal_views[k]))
            res.s_views[k][:] = out + dir.dual_views[k]
        self.comm.allreduce(res.s)

    # the cone sweep of check_cone_points (search.jl:118-136): all owners test their cones at once
    def check_cones(self, cand, irtmu, use_max_prox, taukap_proxsqr, proxsqr_bound):
        ok, agg = 1.0, 0.0
        m = self.model
        for k, proxy in enumerate(m.cones):          # host mirrors of every cone follow the candidate
            np.multiply(cand.primal_views[k], irtmu, out=proxy.point)
            proxy.dual_point[:] = cand.dual_views[k]
            proxy._grad_valid = False
        for k, cone in self._locals():
            cone.load_point(np.ascontiguousarray(cand.primal_views[k]), irtmu)
            cone.load_dual_point(np.ascontiguousarray(cand.dual_views[k]))
            cone.reset_data()
            if ok and cone.is_feas() and cone.is_dual_feas() and cone.check_numerics():
                pk = cone.get_proxsqr(irtmu, use_max_prox)
                agg = max(agg, pk) if use_max_prox else agg + pk
            else:
                ok = 0.0
        flag = np.array([ok])
        self.comm.allreduce(flag, "min")
        val = np.array([agg])
        self.comm.allreduce(val, "max" if use_max_prox else "sum")
        if flag[0] < 0.5:
            return False, 0.0
        total = max(taukap_proxsqr, val[0]) if use_max_prox else taukap_proxsqr + val[0]
        return bool(total < proxsqr_bound), total


# ---------------------------------------------------------------------------------------------------
# distributed versions of the two preprocessing steps that touch all of G
# ---------------------------------------------------------------------------------------------------
def rescale_data_dist(solver):
    """process.jl:13-60 with the column / cone maxima of G reduced over ranks"""
    if not solver.rescale:
        return False
    model = solver.model
    comm = model.comm
    minval = np.sqrt(EPS)
    Gl = model.G_local
    colmax = np.max(np.abs(Gl), axis=0) if Gl.shape[0] else np.zeros(model.n)
    comm.allreduce(colmax, "max")
    c_scale = np.sqrt(np.maximum(np.maximum(np.abs(model.c), colmax), minval))
    h_scale = np.ones(model.q)
    conemax = np.zeros(len(model.cones))
    rowmax = np.zeros(model.q)
    off = 0
    for k in model.local_ks:
        d = model.cones[k].dimension()
        blk = np.abs(Gl[off:off + d, :])
        conemax[k] = blk.max() if blk.size else 0.0
        rowmax[model.cone_idxs[k]] = blk.max(axis=1)
        off += d
    comm.allreduce(conemax, "max")
    comm.allreduce(rowmax, "max")
    for k, cone in enumerate(model.cones):
        idxs = model.cone_idxs[k]
        if getattr(cone, "is_nonnegative", False):
            h_scale[idxs] = np.sqrt(np.maximum(np.maximum(np.abs(model.h[idxs]), rowmax[idxs]), minval))
        else:
            h_scale[idxs] = np.sqrt(max(minval, float(np.max(np.abs(model.h[idxs]))), conemax[k]))
    solver.c_scale, solver.b_scale, solver.h_scale = c_scale, np.zeros(0), h_scale
    model.c = model.c / c_scale
    model.G_local = np.asfortranarray((Gl / c_scale[None, :]) / h_scale[model.local_rows][:, None])
    model.h = model.h / h_scale
    return True


class _GOnly:
    """matrix-free access to the sharded G before the system solver exists (initial point only)"""

    def __init__(self, model):
        self.model, self.comm = model, model.comm

    def mul(self, x):
        out = np.zeros(self.model.q)
        out[self.model.local_rows] = self.model.G_local @ x
        return self.comm.allreduce(out)

    def mul_t(self, z):
        return self.comm.allreduce(self.model.G_local.T @ z[self.model.local_rows])


def lsqr(op, b, atol=1e-14, btol=1e-14, maxiter=None):
    """Paige & Saunders LSQR for min ||G x - b||: the reference's `init_use_indirect` initial point
    (process.jl:81-95 calls IterativeSolvers.lsqr); only products with G and G' are needed."""
    u = b.copy()
    beta = np.linalg.norm(u)
    n = op.model.n
    x = np.zeros(n)
    if beta == 0:
        return x
    u /= beta
    v = op.mul_t(u)
    alpha = np.linalg.norm(v)
    v /= alpha
    w = v.copy()
    phibar, rhobar = beta, alpha
    bnorm = beta
    anorm2 = 0.0
    maxiter = maxiter or 4 * n
    for _ in range(maxiter):
        u = op.mul(v) - alpha * u
        beta = np.linalg.norm(u)
        if beta > 0:
            u /= beta
        anorm2 += alpha * alpha + beta * beta
        v = op.mul_t(u) - beta * v
        alpha = np.linalg.norm(v)
        if alpha > 0:
            v /= alpha
        rho = np.hypot(rhobar, beta)
        cs, sn = rhobar / rho, beta / rho
        theta = sn * alpha
        rhobar = -cs * alpha
        phi = cs * phibar
        phibar = sn * phibar
        x += (phi / rho) * w
        w = v - (theta / rho) * w
        rnorm = phibar
        arnorm = phibar * alpha * abs(cs)
        if rnorm <= btol * bnorm + atol * np.sqrt(anorm2) * np.linalg.norm(x):
            break
        if arnorm <= atol * np.sqrt(anorm2) * max(rnorm, 1e-300):
            break
    return x


def find_initial_x_dist(solver, init_s):
    """least-squares x for G x = h - s over all ranks (process.jl:64-95, indirect branch)"""
    model = solver.model
    solver.x_keep_idxs = np.arange(model.n)
    return lsqr(_GOnly(model), model.h - init_s)


# ---------------------------------------------------------------------------------------------------
# K-panel sharding of ONE replicated model (third bullet: a single cone -- configs[1] / [2])
# ---------------------------------------------------------------------------------------------------
def kshard_range(nrows, rank, world):
    """rows [r0, r1) of the sqrt-Hessian product that rank `rank` sums into the Schur matrix (the K dimension of outer_prod!,
    qrchol.jl:234); 16-row granularity, the same rule as SysSolver::assemble_lhs"""
    per = ((nrows + world - 1) // world + 15) // 16 * 16
    r0 = min(nrows, per * rank)
    return r0, min(nrows, r0 + per)


class KShardQRCholDenseSystemSolver(QRCholDenseSystemSolver):
    """QRCholDenseSystemSolver on a model that every rank holds in full (one cone: its oracles, the factorization and the
    solves do not shard): the ranks split the K dimension of the Schur product and all-reduce the n x n partial sums
    (hyp_sys_set_kshard).  Everything else is replicated and bitwise identical on all ranks, so the plain driver runs unchanged
    on every rank with no further exchange."""

    def __init__(self, comm):
        super().__init__()
        self.comm = comm

    def load(self, solver):
        import ctypes
        import os
        super().load(solver)
        lib, h, comm = L.lib(), self._h, self.comm
        dist, torch = comm.dist, comm.torch
        self._hyp_comm = init_library_rccl(comm, h)
        self.rccl_in_library = self._hyp_comm is not None
        if not self.rccl_in_library:
            nmp = self.n - self.p
            self._stage = torch.empty(max(nmp * nmp, 1024), dtype=torch.float64, device="cuda")
            stage = self._stage
            ops = {0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MAX, 2: dist.ReduceOp.MIN}

            def _allreduce(user, count, op):
                try:
                    dist.all_reduce(stage[:count], op=ops[op])
                    torch.cuda.synchronize()
                    comm._count("library", count, op)
                    return 0
                except Exception as e:   # never let an exception cross the C boundary
                    print("all-reduce callback failed:", e)
                    return 1

            self._cb = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_int)(_allreduce)
            L.check(lib.hyp_sys_set_comm(h, ctypes.cast(self._cb, ctypes.c_void_p), None, ctypes.c_void_p(stage.data_ptr()), int(stage.numel())),
                    "hyp_sys_set_comm")
        L.check(lib.hyp_sys_set_kshard(h, comm.rank, comm.world), "hyp_sys_set_kshard")
        return self

    def agree_any(self, flag):
        """True on every rank iff `flag` is true on some rank: the ranks run the same deterministic solve, and the one decision
        that depends on a rank's own clock (time_limit) must not let one of them leave the all-reduces alone"""
        v = np.array([1.0 if flag else 0.0])
        self.comm.allreduce(v, "max")
        return bool(v[0] > 0.5)

    def close(self):
        """release the library's communicator (every rank, before the process group goes away)"""
        release_library_rccl(getattr(self, "_h", None), getattr(self, "_hyp_comm", None))
        self._hyp_comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
        sup = getattr(super(), "__del__", None)
        if sup is not None:
            sup()
