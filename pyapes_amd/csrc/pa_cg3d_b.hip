// pa_cg3d_b.hip -- instantiations of the tiled kernel: the two BiCGSTAB phases (kernel and launch helpers:
// pa_cg3d_kernel.h)
#include "pa_cg3d_kernel.h"
#include "pa_cg2d_kernel.h"

template <typename T>
int pa_tile3d_bicg_pv(pa_ctx* c, const DevEq<T>& E, Vec<T> r, Vec<T> p, Vec<T> v, const T* r0, T* pnew, T* vnew,
                      double* partials) {
  // (a pitched solve, bicg_run_t: every array of these phases is the ctx's, rows padded to 128 bytes)
  if (c->coord != PA_COORD_XYZ) return 0;   // (the full p / v phase has no 2-D marching form: generic kernel, once per solve)
  const int mode = c->cg_pitch ? 3 : cg3d_mode<T>(c, E, {r.p, p.p, v.p, r0, pnew, vnew, r.glo, r.ghi, p.glo, p.ghi, v.glo, v.ghi}, true);
  if (!mode) return 0;
  Cg3dArgs<T> A;
  memset(&A, 0, sizeof(A));
  fill_common<T>(c, E, A);
  A.ps1 = c->cg_ps1; A.ps0 = c->cg_ps1 * c->G.n1;
  A.r = r; A.d = p; A.v = v; A.aux = r0; A.out = pnew; A.out2 = vnew; A.partials = partials;
  if (c->fold_b_n > 0) {  // close the previous iteration in this kernel's prologue (next state -> the other slot)
    A.pre_part = c->fold_b_part;
    A.pre_n = c->fold_b_n;
    A.sc_w = c->sc_alt;
  }
  int n = launch_any<T, 5>(c, A, mode);
  if (n > 0 && hipGetLastError() != hipSuccess) { pa_set_err(c, "k_cg3d BiCGSTAB p/v launch failed"); return PA_E_HIP; }
  if (n > 0 && c->fold_b_n > 0) {
    SolverScalars* t = c->sc; c->sc = c->sc_alt; c->sc_alt = t;
    c->fold_b_n = 0;
  }
  return n;
}

// the same phase when p' already exists (k_bicg_x formed it, pa_solver.hip): v' = A p' on the interior set, r0 . v'
template <typename T>
int pa_tile3d_bicg_v(pa_ctx* c, const DevEq<T>& E, Vec<T> p, const T* r0, T* vnew, double* partials) {
  const int mode = c->cg_pitch ? 3 : cg3d_mode<T>(c, E, {p.p, r0, vnew, p.glo, p.ghi}, true, false, true);
  if (!mode) return 0;
  const bool rz = c->coord == PA_COORD_RZ;
  Cg3dArgs<T> A;
  memset(&A, 0, sizeof(A));
  fill_common<T>(c, E, A);
  A.ps1 = c->cg_ps1; A.ps0 = c->cg_ps1 * c->G.n1;
  A.d = p; A.aux = r0; A.out2 = vnew; A.partials = partials;
  if (c->fold_b_n > 0) {  // close the previous iteration in this kernel's prologue (next state -> the other slot)
    A.pre_part = c->fold_b_part;
    A.pre_n = c->fold_b_n;
    A.sc_w = c->sc_alt;
  }
  int n = (mode == 1 || mode == 3 || mode == 4) ? launch_cg2d<T, 8>(c, A, mode == 3) : 0;   // large 2-D meshes: marching kernel
  if (n == 0 && rz) return 0;
  if (n == 0) n = launch_any<T, 8>(c, A, mode);
  if (n > 0 && hipGetLastError() != hipSuccess) { pa_set_err(c, "k_cg3d BiCGSTAB v launch failed"); return PA_E_HIP; }
  if (n > 0 && c->fold_b_n > 0) {
    SolverScalars* t = c->sc; c->sc = c->sc_alt; c->sc_alt = t;
    c->fold_b_n = 0;
  }
  return n;
}

template <typename T>
int pa_tile3d_bicg_st(pa_ctx* c, const DevEq<T>& E, Vec<T> r, Vec<T> v, const T* r0, T* s_out, T* t_out,
                      double* partials) {
  const int mode = c->cg_pitch ? 3 : cg3d_mode<T>(c, E, {r.p, v.p, r0, s_out, t_out, r.glo, r.ghi, v.glo, v.ghi}, true, false, true);
  if (!mode) return 0;
  const bool rz = c->coord == PA_COORD_RZ;
  Cg3dArgs<T> A;
  memset(&A, 0, sizeof(A));
  fill_common<T>(c, E, A);
  A.ps1 = c->cg_ps1; A.ps0 = c->cg_ps1 * c->G.n1;
  A.r = r; A.d = v; A.aux = r0; A.out = s_out; A.out2 = t_out; A.partials = partials;
  if (c->fold_a_n > 0) {  // alpha (and the iteration count) in this kernel's prologue
    A.pre_part = (const double*)c->scr[SCR_PART];
    A.pre_n = c->fold_a_n;
    A.sc_w = c->sc;
  }
  int n = (mode == 1 || mode == 3 || mode == 4) ? launch_cg2d<T, 6>(c, A, mode == 3) : 0;
  if (n == 0 && rz) return 0;
  if (n == 0) n = launch_any<T, 6>(c, A, mode);
  if (n > 0 && hipGetLastError() != hipSuccess) { pa_set_err(c, "k_cg3d BiCGSTAB s/t launch failed"); return PA_E_HIP; }
  return n;
}

template int pa_tile3d_bicg_pv<float>(pa_ctx*, const DevEq<float>&, Vec<float>, Vec<float>, Vec<float>, const float*, float*, float*, double*);
template int pa_tile3d_bicg_pv<double>(pa_ctx*, const DevEq<double>&, Vec<double>, Vec<double>, Vec<double>, const double*, double*, double*, double*);
template int pa_tile3d_bicg_v<float>(pa_ctx*, const DevEq<float>&, Vec<float>, const float*, float*, double*);
template int pa_tile3d_bicg_v<double>(pa_ctx*, const DevEq<double>&, Vec<double>, const double*, double*, double*);
template int pa_tile3d_bicg_st<float>(pa_ctx*, const DevEq<float>&, Vec<float>, Vec<float>, const float*, float*, float*, double*);
template int pa_tile3d_bicg_st<double>(pa_ctx*, const DevEq<double>&, Vec<double>, Vec<double>, const double*, double*, double*, double*);
