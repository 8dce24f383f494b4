// pa_sf.hip -- the single-field operations on a tiled mesh: A x (Laplacian, Div, Laplacian + Div), the
// explicit Euler step, the explicit gradient.  The instruction-heavy ones -- everything with a Div term --
// run on k_sf (pa_sf_kernel.h: wave-autonomous marching, no LDS, no barrier) where rows are whole 16-byte
// vectors; the Laplacian alone and the gradient stay on k_cg3d's phases 2 / 7 (pa_cg3d_kernel.h), which move
// their 2 / 4 passes at the speed of a device copy, as does everything k_sf does not take (odd row lengths,
// unaligned operands, tensor coefficient, 2-D meshes).
#include "pa_sf_kernel.h"

template <typename T>
int pa_tile3d_euler(pa_ctx* c, Vec<T> phi, T* out, int kind, double u, const void* u_field, double nu, double dt, int bcl) {
  DevEq<T> E;
  pa_term t;
  memset(&t, 0, sizeof(t));
  t.kind = PA_OP_LAPLACIAN; t.sign = 1.0;
  pa_build_eq<T>(c, 1, &t, E);
  const int mode = cg3d_mode<T>(c, E, {phi.p, out, u_field, phi.glo, phi.ghi});
  if (!mode) return 0;
  if (kind == PA_OP_DIV_CENTRAL && u_field) return 0;  // needs u at the neighbours: generic kernel
  Cg3dArgs<T> A;
  memset(&A, 0, sizeof(A));
  fill_common<T>(c, E, A);
  fill_h<T>(c, A);
  A.d = phi; A.out = out; A.aux = (const T*)u_field; A.u = (T)u; A.p0 = (T)nu; A.p1 = (T)dt; A.kind = kind;
  {  // the BC fill that follows the step kernel (euler_t) rewrites every face plane that has a BC
    int faces = 0;
    for (int f = 0; f < 6; ++f) faces += (c->G.act[f >> 1] && c->bc[f].type != PA_BC_NONE) ? 1 : 0;
    A.out_all = faces == 2 * c->ndim ? 1 : 0;
  }
  if (bcl) {   // BC on load (k_sf only): the fill values of the face interiors are formed from the stencil's own operands
    if (!A.out_all || !sf_applies<T, 3>(c, A, mode) || kind != PA_OP_DIV_UPWIND) return 0;
    A.bcl_c43 = (T)(4.0 / 3.0);
    A.bcl_c13 = (T)(1.0 / 3.0);
    for (int f = 0; f < 6; ++f) {
      const HostBC& b = c->bc[f];
      if (b.vals || b.type < PA_BC_DIRICHLET || b.type > PA_BC_SYMMETRY) return 0;
      A.bcl_type[f] = b.type;
      if (b.type == PA_BC_DIRICHLET) A.bcl_val[f] = (T)b.value;
      if (b.type == PA_BC_NEUMANN) {   // the additive constant as pa_bc.hip forms it
        T pre = (T)((2.0 / 3.0) * b.value);
        pre = pre * (T)b.dxf;
        pre = pre * ((f & 1) == 0 ? (T)-1 : (T)1);
        A.bcl_val[f] = pre;
      }
    }
  }
  int n = 0;
  if (sf_applies<T, 3>(c, A, mode)) {
    switch (kind) {
      case PA_OP_DIV_CENTRAL: n = launch_sf_any<T, 3, PA_OP_DIV_CENTRAL>(c, A); break;
      case PA_OP_DIV_UPWIND_COMPAT: n = launch_sf_any<T, 3, PA_OP_DIV_UPWIND_COMPAT>(c, A); break;
      case PA_OP_DIV_UPWIND: n = launch_sf_any<T, 3, PA_OP_DIV_UPWIND>(c, A); break;
      default: return 0;
    }
  } else {
    n = launch_any<T, 3>(c, A, mode);
  }
  if (n > 0 && hipGetLastError() != hipSuccess) { pa_set_err(c, "k_cg3d Euler launch failed"); return PA_E_HIP; }
  return n;
}

template <typename T>
int pa_tile3d_aop(pa_ctx* c, const DevEq<T>& E, Vec<T> x, T* y, int interior_only) {
  const int mode = cg3d_mode<T>(c, E, {x.p, y, x.glo, x.ghi}, true, true);
  if (!mode) return 0;
  Cg3dArgs<T> A;
  memset(&A, 0, sizeof(A));
  fill_common<T>(c, E, A);
  A.d = x; A.out = y; A.interior_only = interior_only;
  if (A.lap_off) A.aux = E.t[0].u_f;  // explicit upwind Div with a speed field (null: scalar speed)
  int n = 0;
  // the Laplacian alone stays on k_cg3d: both kernels move it at the speed of a device copy (512^3 fp64
  // 0.416 vs 0.423 ms, fp32 0.192 vs 0.200; a copy: 0.414 / 0.200), as they do the gradient
  if (A.kind != 0 && sf_applies<T, 2>(c, A, mode)) {
    switch (A.kind) {   // Laplacian + Div, or the Div term alone (lap_off)
      case PA_OP_DIV_CENTRAL: n = launch_sf_any<T, 2, PA_OP_DIV_CENTRAL>(c, A); break;
      case PA_OP_DIV_UPWIND_COMPAT: n = launch_sf_any<T, 2, PA_OP_DIV_UPWIND_COMPAT>(c, A); break;
      case PA_OP_DIV_UPWIND: n = launch_sf_any<T, 2, PA_OP_DIV_UPWIND>(c, A); break;
      default: return 0;
    }
  } else {
    n = launch_any<T, 2>(c, A, mode);
  }
  if (n > 0 && hipGetLastError() != hipSuccess) { pa_set_err(c, "k_cg3d A x launch failed"); return PA_E_HIP; }
  return n;
}

// explicit gradient, nd components of ncell each (k_grad): geometry / mode check as for a Laplacian
template <typename T>
int pa_tile3d_grad(pa_ctx* c, Vec<T> x, T* y, int nd) {
  DevEq<T> E;
  pa_term t;
  memset(&t, 0, sizeof(t));
  t.kind = PA_OP_LAPLACIAN; t.sign = 1.0;
  pa_build_eq<T>(c, 1, &t, E);
  const int mode = cg3d_mode<T>(c, E, {x.p, y, x.glo, x.ghi});
  if (!mode || nd != c->ndim) return 0;
  Cg3dArgs<T> A;
  memset(&A, 0, sizeof(A));
  fill_common<T>(c, E, A);
  A.grd = E.grd;
  A.gnd = nd;
  A.d = x; A.out = y;
  int n = launch_any<T, 7>(c, A, mode);
  if (n > 0 && hipGetLastError() != hipSuccess) { pa_set_err(c, "k_cg3d grad launch failed"); return PA_E_HIP; }
  return n;
}

template int pa_tile3d_euler<float>(pa_ctx*, Vec<float>, float*, int, double, const void*, double, double, int);
template int pa_tile3d_euler<double>(pa_ctx*, Vec<double>, double*, int, double, const void*, double, double, int);
template int pa_tile3d_grad<float>(pa_ctx*, Vec<float>, float*, int);
template int pa_tile3d_grad<double>(pa_ctx*, Vec<double>, double*, int);
template int pa_tile3d_aop<float>(pa_ctx*, const DevEq<float>&, Vec<float>, float*, int);
template int pa_tile3d_aop<double>(pa_ctx*, const DevEq<double>&, Vec<double>, double*, int);
