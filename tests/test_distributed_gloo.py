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
