"""Compare the device get_directions with the host-composed one on the same right-hand sides (debug aid)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import hypatia_jl_amd as H
from hypatia_jl_amd import solvers as HS
from oracle import instances as I

inst = I.psd_blocks(300, [30, 11, 20], seed=7)
hs = H.Solver(iter_limit=4)
hs.load(H.make_model(inst)); hs.solve()
st, sysv = hs.stepper, hs.syssolver
sysv.update_lhs(hs)
print("cutoff", hs.res_norm_cutoff, "max_ref", hs.max_ref_steps, "mu", hs.mu)
for upd in (HS.update_rhs_cent, HS.update_rhs_pred):
    upd(hs, st.rhs)
    for native in (True, False):
        sysv.native_directions = native
        n0 = hs.n_solves
        hs.worst_dir_res = 0.0
        HS.get_directions(st, hs)
        d = st.dir.vec.copy()
        HS.apply_lhs(st, hs)
        r = st.temp.vec - st.rhs.vec
        print(upd.__name__, "native" if native else "host  ", "solves", hs.n_solves - n0, "reported res", hs.worst_dir_res,
              "true res", np.max(np.abs(r)), "tau/kap res", r[st.rhs.tau_idx], r[-1], "|x|", np.max(np.abs(r[:hs.model.n])),
              "|z|", np.max(np.abs(r[hs.model.n + hs.model.p:st.rhs.tau_idx])), "|s|", np.max(np.abs(r[st.rhs.tau_idx + 1:-1])))
    sysv.native_directions = True
