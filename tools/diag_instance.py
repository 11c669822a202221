"""Run one known-answer instance through the HIP path (verbose), optionally with the host-composed
direction / search routines: python tools/diag_instance.py NAME [reduce=1] [native=1]"""
import sys
sys.path.insert(0, ".")
import hypatia_jl_amd as H
from oracle import instances as I

name = sys.argv[1]
reduce = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
native = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
inst = I.KNOWN_ANSWER[name]()
s = H.Solver(verbose=True, default_tol_relax=10, reduce=reduce)
s.load(H.make_model(inst))
if not native:
    _orig_setup = s.setup
s.solve() if native else None
if not native:
    s2 = H.Solver(verbose=True, default_tol_relax=10, reduce=reduce)
    s2.load(H.make_model(inst))
    s2.setup()
    s2.syssolver.native_directions = False
    while s2.iterate():
        pass
    s = s2
print("status", s.get_status(), "iters", s.get_num_iters(), "worst_dir_res", s.worst_dir_res)
