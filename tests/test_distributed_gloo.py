"""world_size-2 test of the cone-sharded multi-GPU path on CPU (gloo): the sharded driver, with the
oracle injected as the local back end, reproduces the single-process oracle solve."""
import multiprocessing as mp
import os
import socket
import tempfile

import numpy as np
import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_partition_cones():
    from hypatia_jl_amd.distributed import partition_cones
    assert partition_cones(64, 8) == [r for r in range(8) for _ in range(8)]
    assert partition_cones(3, 2) == [0, 0, 1]
    assert partition_cones(1, 4) == [0]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2])
def test_sharded_solve_matches_single_process(world):
    import dist_worker
    from oracle import instances as I
    from oracle.build import make_model
    from oracle.solvers import Solver as OSolver
    inst_args = (40, [6, 5, 4], 3)
    port = _free_port()
    out = os.path.join(tempfile.mkdtemp(), "dist_out.npz")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=dist_worker.run, args=(r, world, port, inst_args, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(280)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    res = np.load(out)
    ref = OSolver(verbose=False)
    ref.load(make_model(I.psd_blocks(*inst_args)))
    ref.solve()
    assert str(res["status"]) == ref.status == "Optimal"
    assert abs(int(res["iters"]) - ref.num_iters) <= 1
    assert abs(float(res["p_obj"]) - ref.primal_obj) <= 1e-7 * (1 + abs(ref.primal_obj))
    assert np.allclose(res["x"], ref.get_x(), rtol=1e-5, atol=1e-7)
    assert np.allclose(res["s"], ref.get_s(), rtol=1e-4, atol=1e-6)
    assert int(res["ncoll"]) > 0


def test_kshard_ranges_tile_the_rows():
    from hypatia_jl_amd.distributed import kshard_range
    for nrows in (1, 15, 16, 17, 210, 20100, 250001):
        for world in (1, 2, 3, 8):
            rs = [kshard_range(nrows, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == nrows
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            assert all((r0 % 16 == 0 or r0 == nrows) and r0 <= r1 for r0, r1 in rs)


@pytest.mark.timeout(300)
def test_kshard_schur_sum_matches_full_assembly():
    """world-2 gloo: the K-panel split of outer_prod! (qrchol.jl:234) for a single-cone model -- partial Gram matrices over
    kshard_range rows, one all-reduce -- against the oracle's full sqrt-Hessian assembly"""
    import dist_worker
    port = _free_port()
    out = os.path.join(tempfile.mkdtemp(), "kshard.npz")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=dist_worker.run_kshard, args=(r, 2, port, (30, [9], 5), out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(280)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    res = np.load(out)
    assert np.allclose(res["lhs"], res["full"], rtol=1e-13, atol=1e-13)
    assert res["ranges"].tolist() == [[0, 32], [32, 45]]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("fail,bad_rank", [("none", 0), ("unique_id", 0), ("init_rank", 1), ("init_rank", 0), ("self_check", 1)])
def test_library_communicator_bringup_is_agreed_on_by_all_ranks(fail, bad_rank):
    """init_library_rccl (the start-up of RCCL inside the library) at world 2 with a stand-in for the hyp_comm_* entry points that
    fails on ONE rank at a chosen step: both ranks must return together -- both with a communicator attached to their solver, or
    both without (falling back to the callback transport), the rank whose own step succeeded having destroyed what it created;
    nobody raises alone, nobody waits in a collective the other never enters (the test would time out)."""
    import dist_worker
    port = _free_port()
    out_dir = tempfile.mkdtemp()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=dist_worker.run_rccl_bringup, args=(r, 2, port, fail, bad_rank, out_dir)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    res = [np.load(os.path.join(out_dir, "bringup_%d.npz" % r)) for r in range(2)]
    want = (fail == "none")
    for r in range(2):
        assert bool(res[r]["got"]) == want and bool(res[r]["attached"]) == want
    if fail in ("init_rank", "self_check"):
        # a rank whose hyp_comm_init_rank succeeded gives the communicator back when the ranks agree to fall back
        for r in range(2):
            created = not (fail == "init_rank" and r == bad_rank)
            assert int(res[r]["destroyed"]) == (1 if created else 0)
