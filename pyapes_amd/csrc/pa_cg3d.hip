// pa_cg3d.hip -- instantiations of the tiled kernel: CG phases A / B, Jacobi sweep
// (kernel and launch helpers: pa_cg3d_kernel.h)
#include "pa_cg3d_kernel.h"
#include "pa_cg2d_kernel.h"

template <typename T>
int pa_cg3d_phase_a(pa_ctx* c, const DevEq<T>& E, Vec<T> r, Vec<T> d, T* dnew, double* partials) {
  const int mode = c->cg_pitch ? 3 : cg3d_mode<T>(c, E, {r.p, d.p, dnew, r.glo, r.ghi, d.glo, d.ghi}, false, false, true);
  if (!mode) return 0;
  const bool rz = c->coord == PA_COORD_RZ;   // axisymmetric: k_cg2d<..., RZ> or nothing (no other tiled kernel has r rows)
  Cg3dArgs<T> A;
  memset(&A, 0, sizeof(A));
  fill_common<T>(c, E, A);
  A.ps1 = c->cg_ps1; A.ps0 = c->cg_ps1 * c->G.n1;
  A.r = r; A.d = d; A.dnew = dnew; A.partials = partials;
  A.reverse = 0;
  if (c->fold_b_n > 0) {  // close the previous iteration in this kernel's prologue (next state -> the other slot)
    A.pre_part = c->fold_b_part;
    A.pre_n = c->fold_b_n;
    A.pre_shell = c->fold_b_shell ? c->fold_b_shell : (const double*)c->scr[SCR_PART2];
    A.pre_nsh = c->fold_b_nsh;
    A.sc_w = c->sc_alt;
    A.pre_sums = pa_sums(c);
  }
  int n = (mode == 1 || mode == 3 || mode == 4) ? launch_cg2d<T, 0>(c, A, mode == 3) : 0;   // 2-D meshes: marching along the slow axis
  if (n == 0 && rz) return 0;
  if (n == 0) n = launch_any<T, 0>(c, A, mode);
  if (n > 0 && hipGetLastError() != hipSuccess) { pa_set_err(c, "k_cg3d phase A launch failed"); return PA_E_HIP; }
  if (n > 0 && c->fold_b_n > 0) {
    SolverScalars* t = c->sc; c->sc = c->sc_alt; c->sc_alt = t;
    c->fold_b_n = c->fold_b_nsh = 0;
    c->fold_b_shell = nullptr;
  }
  return n;
}

template <typename T>
int pa_cg3d_phase_b(pa_ctx* c, const DevEq<T>& E, Vec<T> d, T* x, T* r, double* partials) {
  // the new residual goes where the old one came from -- or, when the placement search moves r (pa_place.hip), straight
  // into its new block: the caller re-points SCR_R after this launch
  T* const r_out = c->cg_r_out ? (T*)c->cg_r_out : r;
  const int mode = c->cg_pitch ? 3 : cg3d_mode<T>(c, E, {d.p, x, r, r_out, d.glo, d.ghi, c->r_send_lo, c->r_send_hi}, false, false, true);
  if (!mode) return 0;
  const bool rz = c->coord == PA_COORD_RZ;
  Cg3dArgs<T> A;
  memset(&A, 0, sizeof(A));
  fill_common<T>(c, E, A);
  A.ps1 = c->cg_ps1; A.ps0 = c->cg_ps1 * c->G.n1;
  A.d = d; A.x = x; A.rw = r; A.rw_out = r_out; A.partials = partials;
  if (!c->slab_fold_live) {   // folded slab iterations: the mid kernel has produced the send planes already
    A.send_lo = (T*)c->r_send_lo; A.send_hi = (T*)c->r_send_hi;
  }
  A.reverse = 1;
  if (c->fold_a_n > 0) {  // alpha of this iteration in this kernel's prologue
    A.pre_part = (const double*)c->scr[SCR_PART] + 2 * (size_t)PA_MAX_PARTIALS;
    A.pre_n = c->fold_a_n;
    A.sc_w = c->sc;
    A.pre_sums = pa_sums(c);
  }
  int n = (mode == 1 || mode == 3 || mode == 4) ? launch_cg2d<T, 1>(c, A, mode == 3) : 0;
  if (n == 0 && rz) return 0;
  if (n == 0) n = launch_any<T, 1>(c, A, mode);
  if (n > 0 && hipGetLastError() != hipSuccess) { pa_set_err(c, "k_cg3d phase B launch failed"); return PA_E_HIP; }
  if (n > 0) c->fold_a_n = 0;
  return n;
}

// ---- Jacobi sweep ---------------------------------------------------------------------------------
template <typename T>
int pa_tile3d_jacobi(pa_ctx* c, const DevEq<T>& E, Vec<T> x, const T* rhs, T* xnew, double omega, double* partials) {
  const int mode = cg3d_mode<T>(c, E, {x.p, rhs, xnew, x.glo, x.ghi}, false, false, true);
  if (!mode) return 0;
  const bool rz = c->coord == PA_COORD_RZ;
  Cg3dArgs<T> A;
  memset(&A, 0, sizeof(A));
  fill_common<T>(c, E, A);
  A.d = x; A.out = xnew; A.aux = rhs; A.p0 = (T)omega; A.partials = partials;
  if (c->fold_b_n > 0) {  // stop test + iteration count of the previous sweep in this kernel's prologue
    A.pre_part = c->fold_b_part;
    A.pre_n = c->fold_b_n;
    A.pre_shell = c->fold_b_shell ? c->fold_b_shell : (const double*)c->scr[SCR_PART2];
    A.pre_nsh = c->fold_b_nsh;
    A.sc_w = c->sc_alt;
    A.pre_sums = pa_sums(c);
  }
  int n = 0;
  if (mode == 1 || mode == 4) {   // large 2-D meshes: marching along the slow axis, consecutive sweeps in opposite directions too
    const bool back2 = c->jac_alt && c->jac_dir && !c->plan_only;
    n = back2 ? launch_cg2d<T, 9>(c, A, false) : launch_cg2d<T, 4>(c, A, false);
    if (n > 0 && !c->plan_only) c->jac_dir ^= 1;
  }
  if (n == 0 && rz) return 0;
  if (n == 0) {   // 3-D: consecutive sweeps march in opposite directions (k_cg3d PHASE 9; option "jac_alt" 0: all forwards)
    const bool back = c->jac_alt && c->jac_dir && !c->plan_only;
    n = back ? launch_any<T, 9>(c, A, mode) : launch_any<T, 4>(c, A, mode);
    if (n > 0 && !c->plan_only) c->jac_dir ^= 1;
  }
  if (n > 0 && hipGetLastError() != hipSuccess) { pa_set_err(c, "k_cg3d Jacobi launch failed"); return PA_E_HIP; }
  if (n > 0 && c->fold_b_n > 0) {
    SolverScalars* t = c->sc; c->sc = c->sc_alt; c->sc_alt = t;
    c->fold_b_n = c->fold_b_nsh = 0;
    c->fold_b_shell = nullptr;
  }
  return n;
}

template int pa_tile3d_jacobi<float>(pa_ctx*, const DevEq<float>&, Vec<float>, const float*, float*, double, double*);
template int pa_tile3d_jacobi<double>(pa_ctx*, const DevEq<double>&, Vec<double>, const double*, double*, double, double*);
template int pa_cg3d_phase_a<float>(pa_ctx*, const DevEq<float>&, Vec<float>, Vec<float>, float*, double*);
template int pa_cg3d_phase_a<double>(pa_ctx*, const DevEq<double>&, Vec<double>, Vec<double>, double*, double*);
template int pa_cg3d_phase_b<float>(pa_ctx*, const DevEq<float>&, Vec<float>, float*, float*, double*);
template int pa_cg3d_phase_b<double>(pa_ctx*, const DevEq<double>&, Vec<double>, double*, double*, double*);
