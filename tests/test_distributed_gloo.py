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


def run_ranks(target, args_of_rank, world, timeout):
    """start `world` spawned processes target(*args_of_rank(rank, port)), wait for them together; a rank that dies takes the others
    down at once (they would wait for it until the timeout and, on a GPU, keep the device busy for every test behind), stragglers
    are terminated at the timeout, and a failed attempt is repeated ONCE on another port (the port found free by _free_port can be
    taken by the time the store binds it: EADDRINUSE killed a whole GPU batch once).  Returns the exit codes of the last attempt."""
    import multiprocessing as mp_
    import time as time_
    codes = []
    for attempt in range(2):
        port = _free_port()
        ctx = mp_.get_context("spawn")
        procs = [ctx.Process(target=target, args=args_of_rank(r, port)) for r in range(world)]
        for p_ in procs:
            p_.start()
        t0 = time_.time()
        while time_.time() - t0 < timeout:
            if all(p_.exitcode is not None for p_ in procs) or any(p_.exitcode not in (None, 0) for p_ in procs):
                break
            time_.sleep(0.2)
        for p_ in procs:
            if p_.exitcode is None:
                p_.join(5 if any(q_.exitcode not in (None, 0) for q_ in procs) else max(0.0, timeout - (time_.time() - t0)))
        for p_ in procs:
            if p_.exitcode is None:
                p_.terminate()
                p_.join(10)
        codes = [p_.exitcode for p_ in procs]
        if all(c == 0 for c in codes):
            break
    return codes


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
    out = os.path.join(tempfile.mkdtemp(), "dist_out.npz")
    codes = run_ranks(dist_worker.run, lambda r, port: (r, world, port, inst_args, out), world, 280)
    assert all(c == 0 for c in codes), codes
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
    out = os.path.join(tempfile.mkdtemp(), "kshard.npz")
    codes = run_ranks(dist_worker.run_kshard, lambda r, port: (r, 2, port, (30, [9], 5), out), 2, 280)
    assert all(c == 0 for c in codes), codes
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
    out_dir = tempfile.mkdtemp()
    codes = run_ranks(dist_worker.run_rccl_bringup, lambda r, port: (r, 2, port, fail, bad_rank, out_dir), 2, 120)
    assert all(c == 0 for c in codes), codes
    res = [np.load(os.path.join(out_dir, "bringup_%d.npz" % r)) for r in range(2)]
    want = (fail == "none")
    for r in range(2):
        assert bool(res[r]["got"]) == want and bool(res[r]["attached"]) == want
    if fail in ("init_rank", "self_check"):
        # a rank whose hyp_comm_init_rank succeeded gives the communicator back when the ranks agree to fall back
        for r in range(2):
            created = not (fail == "init_rank" and r == bad_rank)
            assert int(res[r]["destroyed"]) == (1 if created else 0)
