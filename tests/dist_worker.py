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
        solver.solve()
        if rank == 0:
            np.savez(out_path, status=solver.status, iters=solver.num_iters, p_obj=solver.primal_obj, d_obj=solver.dual_obj,
                     x=solver.get_x(), s=solver.get_s(), z=solver.get_z(), ncoll=comm.n_collectives,
                     hooked=bool(getattr(solver.syssolver, "_hooked", False)), worst_dir_res=solver.worst_dir_res,
                     rccl_in_library=bool(getattr(solver.syssolver, "rccl_in_library", False)), lib_exchanges=_lib_exchanges(solver))
    finally:
        dist.destroy_process_group()
