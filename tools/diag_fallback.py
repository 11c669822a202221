import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hypatia_jl_amd as H
from oracle import instances as I
from oracle.build import make_model as omodel
from oracle.solvers import Solver as OSolver
inst = I.psd_blocks(30, [6, 4], seed=1)
s = H.Solver(verbose=True)
s.iter_callback = lambda sv: print("   fallback", sv.syssolver.used_fallback, "info", sv.syssolver.last_info, "step", getattr(sv.stepper, "cent_only", None), getattr(sv.stepper, "unadj_only", None))
s.load(H.make_model(inst)); s.solve()
o = OSolver(verbose=True)
o.iter_callback = lambda sv: print("   fact kind", sv.syssolver.fact.kind if sv.syssolver.fact else None)
o.load(omodel(inst)); o.solve()
