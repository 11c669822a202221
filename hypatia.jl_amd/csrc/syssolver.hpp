// Device-resident QRCholDenseSystemSolver (/root/reference/src/Solvers/systemsolvers/qrchol.jl:104-257).
#pragma once
#include "cones.hpp"

namespace hyp {

struct SysSolver {
  Ctx& ctx;
  int n, p, q, nmp;
  std::vector<Cone*> cones;
  std::vector<int> offs;        // cone k occupies rows [offs[k], offs[k+1]) of z / s
  // matrices (col-major, device)
  DBuf G;       // q x n   model.G (after preprocessing); aliases GQ2 storage when p == 0
  DBuf GQ1;     // q x p   (p > 0 only)
  DBuf GQ2s;    // q x nmp (own storage only when p > 0)
  DBuf Qm;      // n x n   Ap_Q (p > 0 only)
  DBuf Rinv;    // p x p   inverse of Ap_R (upper), computed once at load
  DBuf HGQ2;    // q x nmp
  DBuf lhs;     // nmp x nmp (upper)
  DBuf lhs_fact, dinv, d_info;
  TriSolvePlan tri;   // super-block inverses of lhs_fact for the one-RHS solves
  // vectors
  DBuf QpbxGHbz, Gx, HGx, GQ1x, HGQ1x, tmpn, sol, rhs, tmpq;
  std::vector<int> use_sqrt;
  bool fact_ok = false;

  SysSolver(Ctx& c, int n_, int p_, int q_, const std::vector<Cone*>& cs);
  const double* GQ2() const { return p == 0 ? G.d() : GQ2s.d(); }
  // host pointers; GQ1/GQ2/Q/R may be null when p == 0 (then GQ2 = G, Q = I)
  void load(const double* hG, const double* hGQ1, const double* hGQ2, const double* hQ, const double* hR);
  void block_hess_prod_vec(double* d_out, const double* d_in);                 // qrchol.jl:87-98 on a q-vector
  void update_lhs_fact(int* info, int* used_fallback);                         // qrchol.jl:201-257
  void assemble_lhs();                                                         //   :214-246 (Schur sum over this process's cones)
  void factor_lhs(int* info, int* used_fallback);                              //   :249-250
  void tri_solves(double* d_x);                                                // both triangular solves of the potrs
  void potrs(double* d_x);                                                     // x <- lhs^-1 x with the current factor (:66-69)
  void solve3(double* d_sol, const double* d_rhs);                             // qrchol.jl:39-85
};

}  // namespace hyp
