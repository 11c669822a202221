"""Worker for tests/test_distributed_gloo.py: runs the cone-sharded driver on CPU with the numpy oracle
injected as the local numerical back end (the product's default back end is the HIP library)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


class OracleLocalSys:
    def __init__(self, comm, model):
        from oracle import linalg as la
        self.la = la
        self.comm, self.model, self.n = comm, model, model.n
        self.cones = [model.cones[k].local for k in model.local_ks]
        self.G = model.G_local
        self.offs = np.cumsum([0] + [c.dimension() for c in self.cones])
        self.lhs = np.zeros((self.n, self.n), order="F")

    def assemble_lhs(self):
        blocks = []
        for i, c in enumerate(self.cones):
            Gk = self.G[self.offs[i]:self.offs[i + 1], :]
            out = np.zeros(Gk.shape, order="F")
            c.sqrt_hess_prod(out, np.asfortranarray(Gk))
            blocks.append(out)
        if blocks:
            H = np.vstack(blocks)
            self.lhs[:] = np.triu(H.T @ H)
        else:
            self.lhs[:] = 0

    def allreduce_lhs(self):
        self.comm.allreduce(self.lhs)

    def factor_lhs(self):
        self.fact = self.la.posdef_fact_copy(self.lhs)
        return (0 if self.fact.success else 1), False

    def potrs(self, x):
        x[:] = self.fact.solve(x)
        return x

    def mul_G(self, trans, x, out):
        out[:] = (self.G.T @ x) if trans else (self.G @ x)
        return out

    def block_hess_prod(self, out_local, in_local):
        for i, c in enumerate(self.cones):
            c.hess_prod(out_local[self.offs[i]:self.offs[i + 1]], in_local[self.offs[i]:self.offs[i + 1]])
        return out_local


def _lib_exchanges(solver):
    try:
        import hypatia_jl_amd as H
        cs = np.zeros(2)
        H._lib.check(H._lib.lib().hyp_sys_comm_stats(solver.syssolver.local._h, H._lib.vec_ptr(cs)), "hyp_sys_comm_stats")
        return cs
    except Exception:
        return np.zeros(2)


def _screen_stats(solver):
    try:
        import ctypes
        import hypatia_jl_amd as H
        u, a, b = ctypes.c_int(0), ctypes.c_longlong(0), ctypes.c_longlong(0)
        H._lib.check(H._lib.lib().hyp_sys_search_screen_stats(solver.syssolver.local._h, ctypes.byref(u), ctypes.byref(a), ctypes.byref(b)),
                     "hyp_sys_search_screen_stats")
        return np.array([u.value, a.value, b.value], dtype=np.float64)
    except Exception:
        return np.zeros(3)


def run(rank, world, port, inst_args, out_path, backend="oracle", transport="gloo"):
    if backend == "hip":
        import torch   # noqa: F401  (first: one HIP runtime per process)
        if transport == "nccl":
            torch.cuda.set_device(0)
    import torch.distributed as dist
    dist.init_process_group(transport, init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        import hypatia_jl_amd as H
        from hypatia_jl_amd import distributed as D
        from oracle import instances as I
        from oracle.build import make_cone
        comm = D.Comm(device="cuda" if backend == "hip" else None)
        inst = I.psd_blocks(*inst_args)
        c, A, b, G, h, specs = inst[:6]
        owners = D.partition_cones(len(specs), world)
        cones, rows, off = [], [], 0
        for k, spec in enumerate(specs):
            mk = H.make_cone if backend == "hip" else make_cone
            local = mk(spec) if owners[k] == rank else None
            dim = spec[1]
            side = int(round((np.sqrt(1 + 8 * dim) - 1) / 2))
            cones.append(D.ShardedCone(comm, owners[k], local, dim, side))
            if owners[k] == rank:
                rows.append(np.arange(off, off + dim))
            off += dim
        rows = np.concatenate(rows) if rows else np.zeros(0, dtype=int)
        model = D.DistModel(comm, c, h, G[rows, :], cones, owners)
        lb = D.HipLocalSys if backend == "hip" else OracleLocalSys
        solver = H.Solver(verbose=False, syssolver=D.DistQRCholDenseSystemSolver(comm, local_backend=lb))
        solver.load(model)
        marks = []
        solver.iter_callback = lambda sv: marks.append(len(comm.payload_log))   # (first call: set-up is over; last: before postprocess)
        solver.solve()
        in_loop = comm.payload_log[marks[0]:marks[-1]] if len(marks) >= 2 else []
        if rank == 0:
            np.savez(out_path, status=solver.status, iters=solver.num_iters, p_obj=solver.primal_obj, d_obj=solver.dual_obj,
                     x=solver.get_x(), s=solver.get_s(), z=solver.get_z(), ncoll=comm.n_collectives,
                     max_host_payload_in_loop=(max(in_loop) if in_loop else 0), host_collectives_in_loop=len(in_loop),
                     row_local=bool(getattr(solver.syssolver, "row_local", False)), q=model.q, n=model.n,
                     hooked=bool(getattr(solver.syssolver, "_hooked", False)), worst_dir_res=solver.worst_dir_res,
                     rccl_in_library=bool(getattr(solver.syssolver, "rccl_in_library", False)), lib_exchanges=_lib_exchanges(solver),
                     screen_stats=_screen_stats(solver), n_solves=solver.n_solves,
                     comm_hist=np.array(solver.syssolver.comm_hist() if hasattr(solver.syssolver, "comm_hist") and
                                        getattr(solver.syssolver, "_hooked", False) else [0] * 16))
    finally:
        dist.destroy_process_group()


def run_kshard(rank, world, port, inst_args, out_path, backend="oracle"):
    """K-panel sharding of ONE replicated model (hypatia.jl_amd.distributed.kshard_range / KShardQRCholDenseSystemSolver).
    oracle: the partition rule on the CPU -- each rank sums its rows of the oracle's sqrt-Hessian product into the Schur
    matrix, one all-reduce, compared by the parent with the oracle's full assembly.  hip: the whole solve on the GPU."""
    if backend == "hip":
        import torch   # noqa: F401
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        import hypatia_jl_amd as H
        from hypatia_jl_amd import distributed as D
        from oracle import instances as I
        if inst_args and inst_args[0] == "polymin":
            inst = I.polymin(*inst_args[1:])
        else:
            inst = I.psd_blocks(*inst_args)
        if backend == "hip":
            comm = D.Comm(device="cuda")
            solver = H.Solver(verbose=False, syssolver=D.KShardQRCholDenseSystemSolver(comm))
            solver.load(H.make_model(inst))
            solver.solve()
            cs = _lib_exchanges_plain(solver)
            if rank == 0:
                np.savez(out_path, status=solver.status, iters=solver.num_iters, p_obj=solver.primal_obj, x=solver.get_x(), exchanges=cs,
                         ncoll=comm.n_collectives)
            return
        from oracle.build import make_cone
        comm = D.Comm()
        c, A, b, G, h, specs = inst[:6]
        assert len(specs) == 1
        cone = make_cone(specs[0]).setup_data()
        pt = np.zeros(cone.dimension())
        cone.set_initial_point(pt)
        pt += 0.05 * np.random.default_rng(7).standard_normal(pt.shape) / np.sqrt(pt.shape[0])
        cone.load_point(pt, 0.9)
        assert cone.is_feas()
        HG = np.zeros(G.shape, order="F")
        cone.sqrt_hess_prod(HG, np.asfortranarray(G))
        r0, r1 = D.kshard_range(G.shape[0], rank, world)
        part = HG[r0:r1].T @ HG[r0:r1]
        comm.allreduce(part)
        if rank == 0:
            np.savez(out_path, lhs=part, full=HG.T @ HG, ranges=np.array([D.kshard_range(G.shape[0], r, world) for r in range(world)]))
    finally:
        dist.destroy_process_group()


def _lib_exchanges_plain(solver):
    import hypatia_jl_amd as H
    cs = np.zeros(2)
    H._lib.check(H._lib.lib().hyp_sys_comm_stats(solver.syssolver._h, H._lib.vec_ptr(cs)), "hyp_sys_comm_stats")
    return cs


def run_comm_selftest():
    import ctypes
    import torch
    torch.cuda.set_device(0)
    t = torch.arange(1000, dtype=torch.float64, device="cuda")
    import hypatia_jl_amd as H
    L = H._lib
    lib = L.lib()
    uid = ctypes.create_string_buffer(128)
    L.check(lib.hyp_comm_unique_id(uid), "hyp_comm_unique_id")
    assert any(b != 0 for b in uid.raw)
    hc = ctypes.c_void_p()
    L.check(lib.hyp_comm_init_rank(L.ctx(), 1, 0, uid, ctypes.byref(hc)), "hyp_comm_init_rank")
    for op in (0, 1, 2):
        L.check(lib.hyp_comm_allreduce(hc, ctypes.c_void_p(t.data_ptr()), 1000, op), "hyp_comm_allreduce")
    assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float64))
    L.check(lib.hyp_comm_destroy(hc), "hyp_comm_destroy")


class FakeCommLib:
    """stand-in for the library's communicator entry points (hyp_comm_*): fails where `fail` says, on the rank `bad_rank`; its
    all-reduce is torch.distributed's on the same buffer, so that a healthy run passes the self-check"""

    def __init__(self, rank, world, fail, bad_rank, dist, torch):
        self.rank, self.world, self.fail, self.bad, self.dist, self.torch = rank, world, fail, bad_rank, dist, torch
        self.destroyed = 0
        self.attached = False

    def _bad(self, what):
        return self.fail == what and self.rank == self.bad

    def hyp_comm_unique_id(self, buf):
        buf.raw = bytes(range(128))
        return 1 if self._bad("unique_id") else 0

    def hyp_comm_init_rank(self, ctx, world, rank, uid, out):
        import ctypes
        if self._bad("init_rank"):
            return 1
        assert bytes(uid.raw) == bytes(range(128)) and world == self.world and rank == self.rank
        ctypes.cast(out, ctypes.POINTER(ctypes.c_void_p))[0] = 4242
        return 0

    def hyp_comm_allreduce(self, hc, ptr, count, op):
        import ctypes
        import numpy as np
        arr = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_double)), shape=(count,))
        t = self.torch.from_numpy(arr)
        self.dist.all_reduce(t)
        if self._bad("self_check"):
            arr[0] += 1.0          # a communicator that returns a wrong sum on one rank
        return 0

    def hyp_comm_destroy(self, hc):
        self.destroyed += 1
        return 0

    def hyp_sys_set_comm_rccl(self, sys_handle, hc):
        self.attached = True
        return 0


def run_rccl_bringup(rank, world, port, fail, bad_rank, out_dir):
    """hypatia.jl_amd.distributed.init_library_rccl over gloo with the stand-in library: every rank must come out the same way"""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        from hypatia_jl_amd import distributed as D
        comm = D.Comm(device="cpu")
        lib = FakeCommLib(rank, world, fail, bad_rank, dist, torch)
        hc = D.init_library_rccl(comm, None, lib=lib, lib_ctx=None)
        np.savez(os.path.join(out_dir, "bringup_%d.npz" % rank), got=(hc is not None), attached=lib.attached, destroyed=lib.destroyed)
    finally:
        dist.destroy_process_group()
