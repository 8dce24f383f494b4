// pa_ops.hip -- generic (any dimension / extent / term list) operator kernels and the explicit entry
// points: y = A x, Laplacian / Grad / Div with edge=True post-passes, rhs adjustment of Solver.set_eq,
// explicit Euler step.  The tiled kernels (pa_cg3d*.hip) take over where they apply.
#include "pa_host.h"

#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>

// ---- y = A(x) (pyapes/solver/ops.py:122-154) -----------------------------------------
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_aop(DevGeom G, DevEq<T> E, Vec<T> xv, T* __restrict__ y,
                                                   int interior_only) {
  FieldAcc<T> acc{xv};
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < G.ncell;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t i, j, k;
    pa_decode(G, idx, i, j, k);
    T out = (T)0;
    if (!interior_only || pa_in_S(G, i, j, k)) {
      T xc = xv.p[idx];
      out = pa_apply_terms<T>(G, E, acc, i, j, k, xc);
    }
    y[idx] = out;
  }
}

// ---- explicit gradient: y[(a), n...] (fdc.py:80-87) -----------------------------------
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_grad(DevGeom G, DevEq<T> E, Vec<T> xv, T* __restrict__ y,
                                                    int nd) {
  FieldAcc<T> acc{xv};
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < G.ncell;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t i, j, k;
    pa_decode(G, idx, i, j, k);
    int64_t g[3], N[3];
    pa_gidx(G, i, j, k, g, N);
    T xc = xv.p[idx];
    for (int a = 3 - nd; a < 3; ++a) {
      T cP = E.grd.g[a], cC = (T)0, cM = E.grd.mg[a];
      int rc = pa_row_case(G, a, g[a], N[a], G.treat);
      if (rc == 1) { cP = E.grd.lo_p[a]; cC = E.grd.lo_c[a]; cM = (T)0; }
      if (rc == 2) { cP = (T)0; cC = E.grd.hi_c[a]; cM = E.grd.hi_m[a]; }
      if (G.bct[2 * a] == 4 && g[a] == 1) cM = (T)0;
      if (G.bct[2 * a + 1] == 4 && g[a] == N[a] - 2) cP = (T)0;
      T xp, xm;
      pa_nbrs<T>(G, acc, a, i, j, k, xp, xm);
      T s = cP * xp;
      T m = cC * xc;
      s = s + m;
      m = cM * xm;
      s = s + m;
      y[(int64_t)(a - (3 - nd)) * G.ncell + idx] = s;
    }
  }
}

// ---- edge=True one-sided boundary formulas (fdc.py:203-288) ---------------------------
// mode 0: laplacian (y is one field; the LAST mesh axis whose index is on the boundary wins,
// because the reference overwrites faces axis by axis); mode 1: grad (y[a] on faces normal to a).
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_edge(DevGeom G, DevEq<T> E, const T* __restrict__ x,
                                                    T* __restrict__ y, int nd, int mode, T u = (T)0,
                                                    const T* __restrict__ u_f = nullptr) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < G.ncell;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t i, j, k;
    pa_decode(G, idx, i, j, k);
    int64_t c[3] = {i, j, k};
    int64_t n[3] = {G.n0, G.n1, G.n2};
    int64_t st[3] = {G.s0, G.s1, 1};
    if (mode == 0) {
      int sel = -1;
      for (int a = 3 - nd; a < 3; ++a)
        if (c[a] == 0 || c[a] == n[a] - 1) sel = a;
      if (sel < 0) continue;
      int64_t dir = (c[sel] == 0) ? 1 : -1;
      T v0 = x[idx], v1 = x[idx + dir * st[sel]], v2 = x[idx + 2 * dir * st[sel]],
        v3 = x[idx + 3 * dir * st[sel]];
      T s = (T)2 * v0;
      T m = (T)5 * v1;
      s = s - m;
      m = (T)4 * v2;
      s = s + m;
      s = s - v3;
      T h2 = E.grd.h[sel] * E.grd.h[sel];
      y[idx] = s / h2;
    } else if (mode == 1) {
      for (int a = 3 - nd; a < 3; ++a) {
        if (!(c[a] == 0 || c[a] == n[a] - 1)) continue;
        int64_t dir = (c[a] == 0) ? 1 : -1;
        T v0 = x[idx], v1 = x[idx + dir * st[a]], v2 = x[idx + 2 * dir * st[a]];
        T s = (T)1.5 * v0;
        T m = (T)2 * v1;
        s = s - m;
        m = (T)0.5 * v2;
        s = s + m;
        if (c[a] == 0) s = -s;
        y[(int64_t)(a - (3 - nd)) * G.ncell + idx] = s / E.grd.h[a];
      }
    } else {
      // Div, 1-D (fdc.py:316-348): -+(3/2 v0 - 2 v1 + 1/2 v2) / dx * adv on the two end nodes
      const int a = 2;
      if (!(c[a] == 0 || c[a] == n[a] - 1)) continue;
      int64_t dir = (c[a] == 0) ? 1 : -1;
      T v0 = x[idx], v1 = x[idx + dir], v2 = x[idx + 2 * dir];
      T s = (T)1.5 * v0;
      T m = (T)2 * v1;
      s = s - m;
      m = (T)0.5 * v2;
      s = s + m;
      if (c[a] == 0) s = -s;
      s = s / E.grd.h[a];
      y[idx] = s * (u_f ? u_f[idx] : u);
    }
  }
}

// ---- rhs adjustment of Solver.set_eq (ops.py:63-77; fdc.py:426-458, 505-540, 667-694) --
template <typename T>
struct RhsFace {
  int type;       // PA_BC_*
  T sval;         // scalar V
  const T* vals;  // per-node V or null
};
template <typename T>
struct RhsArgs {
  RhsFace<T> f[6];
  int order[6];   // internal face ids in list order
  int nfaces;
  T c23, c13;     // (T)(2/3), (T)(1/3)
  T h[3];
  // only nodes one step inside a Neumann face are touched: the kernel visits those layers, not the mesh
  int nlay;             // number of layers (<= 6); 0: visit every cell
  int lay_axis[6];      // internal axis of layer w
  int64_t lay_pos[6];   // its LOCAL index along that axis
  int64_t lay_start[7]; // prefix sums of the layer sizes
};

template <typename T>
__device__ __forceinline__ T pa_face_val(const DevGeom& G, const RhsFace<T>& F, int a, int64_t i, int64_t j,
                                         int64_t k) {
  if (!F.vals) return F.sval;
  if (a == 0) return F.vals[j * G.n2 + k];
  if (a == 1) return F.vals[i * G.n2 + k];
  return F.vals[i * G.n1 + j];
}

template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_rhs_adjust(DevGeom G, DevEq<T> E, RhsArgs<T> R,
                                                          T* __restrict__ rhs) {
  const int64_t total = R.nlay ? R.lay_start[R.nlay] : G.ncell;
  for (int64_t tix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; tix < total;
       tix += (int64_t)gridDim.x * blockDim.x) {
    int64_t idx = tix, i, j, k;
    if (R.nlay) {
      // which layer, which node of it; a node on two layers belongs to the first one
      int wsel = 0, ax = 0;
      int64_t q = 0, pos = 0;
#pragma unroll
      for (int w = 0; w < 6; ++w)
        if (w < R.nlay && tix >= R.lay_start[w] && tix < R.lay_start[w + 1]) {
          wsel = w; ax = R.lay_axis[w]; pos = R.lay_pos[w]; q = tix - R.lay_start[w];
        }
      if (ax == 0) { i = pos; j = q / G.n2; k = q - j * G.n2; }
      else if (ax == 1) { j = pos; i = q / G.n2; k = q - i * G.n2; }
      else { k = pos; i = q / G.n1; j = q - i * G.n1; }
      bool dup = false;
#pragma unroll
      for (int w = 0; w < 6; ++w)
        if (w < wsel) {
          const int64_t cw = R.lay_axis[w] == 0 ? i : (R.lay_axis[w] == 1 ? j : k);
          if (cw == R.lay_pos[w]) dup = true;
        }
      if (dup) continue;
      idx = i * G.s0 + j * G.s1 + k;
    } else {
      pa_decode(G, idx, i, j, k);
    }
    int64_t g[3], N[3];
    pa_gidx(G, i, j, k, g, N);
    T val = rhs[idx];
    bool touched = false;
    for (int q = 0; q < E.nterms; ++q) {
      const DevTerm<T>& t = E.t[q];
      T adj = (T)0;
      bool any = false;
      // reference loop nest: for axis j: for bc in list order (only faces normal to j contribute)
      for (int a = 0; a < 3; ++a) {
        if (!G.act[a]) continue;
        for (int w = 0; w < R.nfaces; ++w) {
          int fc = R.order[w];
          if ((fc >> 1) != a) continue;
          if (R.f[fc].type != 2) continue;
          int side = fc & 1;
          int64_t prev = side == 0 ? pa_wrap(1, N[a]) : pa_wrap(N[a] - 2, N[a]);
          if (g[a] != prev) continue;
          T V = pa_face_val<T>(G, R.f[fc], a, i, j, k);
          T nv = side == 0 ? (T)-1 : (T)1;
          T vn = V * nv;
          if (t.kind == 0) {            // laplacian: += (2/3 - alpha)(V n)/h   (fdc.py:440-453)
            T f23 = (E.rz && a == PA_RZ_AXIS) ? E.rz[3 * E.rz_n + g[a]] : R.c23;
            T s = f23 * vn;
            s = s / R.h[a];
            adj = adj + s;
          } else if (t.kind == 1) {     // grad: -= (1/3)(V n) * 1      (fdc.py:526-537)
            T s = R.c13 * vn;
            adj = adj - s;
          } else {                      // div: -= (1/3)(V n) * gamma   (fdc.py:680-686)
            T ucen = t.u_f ? t.u_f[idx] : t.u;
            T gm;
            if (t.kind == 2) gm = (T)2 * ucen;
            else {
              // upwind: lower face uses 2*max(u,0), upper face 2*min(u,0)
              T mx = ucen > (T)0 ? ucen : (T)0, mn = ucen < (T)0 ? ucen : (T)0;
              gm = side == 0 ? (T)2 * mx : (T)2 * mn;
            }
            T s = R.c13 * vn;
            s = s * gm;
            adj = adj - s;
          }
          any = true;
        }
      }
      if (any) { val = val + adj; touched = true; }
    }
    if (touched) rhs[idx] = val;
  }
}


// ---- explicit Euler step [new, SURVEY a15] ----------------------------------------------
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_euler(DevGeom G, DevEq<T> Elap, DevEq<T> Eadv, Vec<T> pv,
                                                     T* __restrict__ out, T nu, T dt) {
  FieldAcc<T> acc{pv};
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < G.ncell;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t i, j, k;
    pa_decode(G, idx, i, j, k);
    T pc = pv.p[idx];
    T v = pc;
    if (pa_in_S(G, i, j, k)) {
      T lap = pa_apply_terms<T>(G, Elap, acc, i, j, k, pc);
      T adv = pa_apply_terms<T>(G, Eadv, acc, i, j, k, pc);
      T a = nu * lap;
      a = a - adv;
      a = dt * a;
      v = pc + a;
    }
    out[idx] = v;
  }
}

// ---- host side ----------------------------------------------------------------------------------
// Grad inside an operator sum only makes sense in 1-D (the reference reshapes the
// (1, mesh.dim, n...) result onto the target, ops.py:145-147)
int pa_check_eq_applicable(pa_ctx* c) {
  for (int q = 0; q < c->nterms; ++q)
    if (c->terms[q].kind == PA_OP_GRAD && c->ndim != 1) {
      pa_set_err(c, "Grad in a solver equation is 1-D only (ops.py:145-147 view)");
      return PA_E_ARG;
    }
  return PA_OK;
}

// -------- typed implementations behind the remaining entry points ------------------------
template <typename T>
static int aop_t(pa_ctx* c, const T* x, T* y, int interior_only, int nterms, const pa_term* terms) {
  DevEq<T> E;
  pa_build_eq<T>(c, nterms, terms, E);
  Vec<T> xv = pa_vec_self<T>(c, x);
  if (c->G.n0 != c->G.g0 && c->ndim == 3) {
    // slab: ghost planes of x must have been supplied
    if (!c->x_glo || !c->x_ghi) { pa_set_err(c, "pa_aop on a slab needs ghost planes (pa_x_ghost_set)"); return PA_E_STATE; }
    xv.glo = (const T*)c->x_glo;
    xv.ghi = (const T*)c->x_ghi;
  }
  int fr = pa_tile3d_aop<T>(c, E, xv, y, interior_only);
  if (fr < 0) return fr;
  if (fr == 0)
    hipLaunchKernelGGL(k_aop<T>, dim3(pa_grid_blocks(c->G.ncell)), dim3(PA_BLOCK), 0, c->stream, c->G, E, xv, y,
                       interior_only);
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

template <typename T>
static int rhs_adjust_t(pa_ctx* c, T* rhs) {
  DevEq<T> E;
  pa_build_eq<T>(c, c->nterms, c->terms, E);
  RhsArgs<T> R;
  memset(&R, 0, sizeof(R));
  bool any = false;
  for (int f = 0; f < 6; ++f) {
    R.f[f].type = c->bc[f].type;
    R.f[f].sval = (T)c->bc[f].value;
    R.f[f].vals = (const T*)c->bc[f].vals;
    if (c->bc[f].type == PA_BC_NEUMANN) any = true;
  }
  R.nfaces = c->nbc;
  for (int w = 0; w < c->nbc; ++w) R.order[w] = c->bc_order[w];
  R.c23 = (T)(2.0 / 3.0);
  R.c13 = (T)(1.0 / 3.0);
  for (int a = 0; a < 3; ++a) R.h[a] = (T)c->dx[a];
  if (!any) return PA_OK;
  // the layers one step inside each Neumann face (global node 1 / N-2 of its axis), as far as this rank owns them
  const DevGeom& G = c->G;
  const int64_t Ng[3] = {G.g0, G.n1, G.n2}, nl[3] = {G.n0, G.n1, G.n2};
  R.nlay = 0;
  R.lay_start[0] = 0;
  for (int f = 0; f < 6; ++f) {
    const int a = f >> 1;
    if (c->bc[f].type != PA_BC_NEUMANN || !G.act[a]) continue;
    int64_t prev = (f & 1) == 0 ? 1 : Ng[a] - 2;
    prev = ((prev % Ng[a]) + Ng[a]) % Ng[a];
    const int64_t pos = a == 0 ? prev - G.off0 : prev;
    if (pos < 0 || pos >= nl[a]) continue;   // another rank's plane
    const int64_t size = a == 0 ? G.n1 * G.n2 : (a == 1 ? G.n0 * G.n2 : G.n0 * G.n1);
    R.lay_axis[R.nlay] = a;
    R.lay_pos[R.nlay] = pos;
    R.lay_start[R.nlay + 1] = R.lay_start[R.nlay] + size;
    ++R.nlay;
  }
  if (R.nlay == 0) return PA_OK;             // no Neumann layer on this rank
  if (c->rhs_full) R.nlay = 0;   // option "rhs_full" (tests: same bits)
  const int64_t work = R.nlay ? R.lay_start[R.nlay] : G.ncell;
  hipLaunchKernelGGL(k_rhs_adjust<T>, dim3(pa_grid_blocks(work)), dim3(PA_BLOCK), 0, c->stream, c->G, E, R, rhs);
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

template <typename T>
static int lap_t(pa_ctx* c, const T* x, T* y, int edge) {
  pa_term t;
  memset(&t, 0, sizeof(t));
  t.kind = PA_OP_LAPLACIAN; t.sign = 1.0; t.has_coeff = 0;
  int rc = aop_t<T>(c, x, y, 0, 1, &t);
  if (rc) return rc;
  if (edge) {
    for (int a = 0; a < c->ndim; ++a) {
      int64_t n = a + (3 - c->ndim) == 0 ? c->G.n0 : (a + (3 - c->ndim) == 1 ? c->G.n1 : c->G.n2);
      if (n < 4) { pa_set_err(c, "edge laplacian needs >= 4 nodes per axis"); return PA_E_ARG; }
    }
    if (c->G.n0 != c->G.g0 && c->ndim == 3) { pa_set_err(c, "edge operators are single-GPU only"); return PA_E_ARG; }
    DevEq<T> E;
    pa_build_eq<T>(c, 1, &t, E);
    hipLaunchKernelGGL(k_edge<T>, dim3(pa_grid_blocks(c->G.ncell)), dim3(PA_BLOCK), 0, c->stream, c->G, E, x, y,
                       c->ndim, 0);
    PA_HIP(c, hipGetLastError());
  }
  return PA_OK;
}

template <typename T>
static int grad_t(pa_ctx* c, const T* x, T* y, int edge) {
  pa_term t;
  memset(&t, 0, sizeof(t));
  t.kind = PA_OP_GRAD; t.sign = 1.0;
  DevEq<T> E;
  pa_build_eq<T>(c, 1, &t, E);
  Vec<T> xv = pa_vec_self<T>(c, x);
  if (c->G.n0 != c->G.g0 && c->ndim == 3) {
    if (!c->x_glo || !c->x_ghi) { pa_set_err(c, "pa_grad on a slab needs ghost planes"); return PA_E_STATE; }
    xv.glo = (const T*)c->x_glo; xv.ghi = (const T*)c->x_ghi;
  }
  int fr = pa_tile3d_grad<T>(c, xv, y, c->ndim);
  if (fr < 0) return fr;
  if (fr == 0)
    hipLaunchKernelGGL(k_grad<T>, dim3(pa_grid_blocks(c->G.ncell)), dim3(PA_BLOCK), 0, c->stream, c->G, E, xv, y,
                       c->ndim);
  if (edge) {
    if (c->G.n0 != c->G.g0 && c->ndim == 3) { pa_set_err(c, "edge operators are single-GPU only"); return PA_E_ARG; }
    hipLaunchKernelGGL(k_edge<T>, dim3(pa_grid_blocks(c->G.ncell)), dim3(PA_BLOCK), 0, c->stream, c->G, E, x, y,
                       c->ndim, 1);
  }
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

static int check_div_kind(pa_ctx* c, int kind) {
  if (kind != PA_OP_DIV_CENTRAL && kind != PA_OP_DIV_UPWIND_COMPAT && kind != PA_OP_DIV_UPWIND) {
    pa_set_err(c, "bad div kind %d", kind);
    return PA_E_ARG;
  }
  if (kind == PA_OP_DIV_CENTRAL)
    for (int f = 0; f < 6; ++f)
      if (c->G.treat[f]) {
        pa_set_err(c, "central Div with neumann/symmetry faces: the reference raises IndexError (fdc.py:583)");
        return PA_E_ARG;
      }
  return PA_OK;
}

template <typename T>
static int euler_t(pa_ctx* c, const T* in, T* out, int kind, double u, const void* u_field, double nu, double dt) {
  pa_term tl, ta;
  memset(&tl, 0, sizeof(tl));
  memset(&ta, 0, sizeof(ta));
  tl.kind = PA_OP_LAPLACIAN; tl.sign = 1.0;
  ta.kind = kind; ta.sign = 1.0; ta.u = u; ta.u_field = u_field;
  DevEq<T> El, Ea;
  pa_build_eq<T>(c, 1, &tl, El);
  pa_build_eq<T>(c, 1, &ta, Ea);
  Vec<T> pv = pa_vec_self<T>(c, in);
  if (c->G.n0 != c->G.g0 && c->ndim == 3) {
    // a slab (pyapes_amd/slab.py SlabEuler): ghost planes from pa_slab_set; a NULL one marks a physical end, whose
    // boundary plane no interior node reads across -- the field's own end plane stands in for the speculative loads
    if (!c->slab) { pa_set_err(c, "pa_euler_step on a slab needs pa_slab_set (ghost planes)"); return PA_E_STATE; }
    pv.glo = c->x_glo ? (const T*)c->x_glo : in;
    pv.ghi = c->x_ghi ? (const T*)c->x_ghi : in + (c->G.n0 - 1) * c->G.s0;
  }
  if (c->profile) (void)hipEventRecord(c->pev[0], c->stream);   // slot 0: the step kernel (without its BC fill)
  int fr = pa_tile3d_euler<T>(c, pv, out, kind, u, u_field, nu, dt);
  if (fr < 0) return fr;
  if (fr == 0)
    hipLaunchKernelGGL(k_euler<T>, dim3(pa_grid_blocks(c->G.ncell)), dim3(PA_BLOCK), 0, c->stream, c->G, El, Ea, pv,
                       out, (T)nu, (T)dt);
  if (c->profile) pa_profile_stop(c, 0);
  PA_HIP(c, hipGetLastError());
  // slab mode: the step kernel alone.  The fill of a periodic axis 0 reads planes of the NEW field that live on the
  // other end rank of the ring, so the driver exchanges those first and then calls pa_apply_bc itself.
  if (c->slab) return PA_OK;
  return pa_bc_apply_auto<T>(c, out, false);
}

// One step of the march in the "BC on load" form (pa_sf_kernel.h): the step kernel alone, no fill behind it -- the
// boundary nodes of `out` stay whatever they were.  1: launched; 0: the form does not apply here; < 0: error.
template <typename T>
static int euler_bcl_t(pa_ctx* c, const T* in, T* out, int kind, double u, const void* u_field, double nu, double dt) {
  Vec<T> pv = pa_vec_self<T>(c, in);
  if (c->profile) (void)hipEventRecord(c->pev[0], c->stream);
  const int fr = pa_tile3d_euler<T>(c, pv, out, kind, u, u_field, nu, dt, 1);
  if (fr <= 0) return fr;
  if (c->profile) pa_profile_stop(c, 0);
  return 1;
}

// ---- vector steps of the host-stepped solver loops (pyapes_amd/solver/host_stepped.py) ------------------------------
// out = y + a x, the product rounded before the sum (torch: y + a * x; "y - a x" is the same bits with -a)
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_vec_axpy(T* __restrict__ out, const T* __restrict__ y, T a,
                                                        const T* __restrict__ x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    T t = a * x[i];
    out[i] = y[i] + t;
  }
}

// partial sums of a.b (diff = 0) or of (a - b)^2 (diff = 1), products rounded in T, summed in double
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_vec_dot(const T* __restrict__ a, const T* __restrict__ b, int diff, int64_t n,
                                                       double* __restrict__ partials) {
  double s[1] = {0.0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    T p;
    if (diff) { T d = a[i] - b[i]; p = d * d; } else { p = a[i] * b[i]; }
    s[0] += (double)p;
  }
  pa_block_reduce_store<1>(s, partials);
}

// x <- 0 off the interior set of the bound BC list (the residual lives on S, linalg.py:99-101)
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_vec_mask_interior(DevGeom G, T* __restrict__ x) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < G.ncell; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t i, j, k;
    pa_decode(G, idx, i, j, k);
    if (!pa_in_S(G, i, j, k)) x[idx] = (T)0;
  }
}

__global__ void __launch_bounds__(PA_BLOCK) k_vec_dot_final(const double* __restrict__ partials, int nblk, double* __restrict__ out) {
  __shared__ double sm[PA_BLOCK / 64];
  double v = 0.0;
  for (int b = threadIdx.x; b < nblk; b += blockDim.x) v += partials[b];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = sm[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t += sm[w];
    out[0] = t;
  }
}

extern "C" {

int pa_vec_axpy(pa_ctx* c, void* out, const void* y, double a, const void* x) {
  if (!c || !c->grid_set || !out || !y || !x) return PA_E_STATE;
  PA_HIP(c, hipSetDevice(c->device));
  const int nb = pa_grid_blocks(c->G.ncell);
  if (c->dtype == PA_F64)
    hipLaunchKernelGGL(k_vec_axpy<double>, dim3(nb), dim3(PA_BLOCK), 0, c->stream, (double*)out, (const double*)y, a,
                       (const double*)x, c->G.ncell);
  else
    hipLaunchKernelGGL(k_vec_axpy<float>, dim3(nb), dim3(PA_BLOCK), 0, c->stream, (float*)out, (const float*)y, (float)a,
                       (const float*)x, c->G.ncell);
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

int pa_vec_mask_interior(pa_ctx* c, void* x) {
  if (!c || !c->grid_set || !x) return PA_E_STATE;
  PA_HIP(c, hipSetDevice(c->device));
  const int nb = pa_grid_blocks(c->G.ncell);
  if (c->dtype == PA_F64) hipLaunchKernelGGL(k_vec_mask_interior<double>, dim3(nb), dim3(PA_BLOCK), 0, c->stream, c->G, (double*)x);
  else hipLaunchKernelGGL(k_vec_mask_interior<float>, dim3(nb), dim3(PA_BLOCK), 0, c->stream, c->G, (float*)x);
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

int pa_vec_dot(pa_ctx* c, const void* a, const void* b, int diff, double* result) {
  if (!c || !c->grid_set || !a || !b || !result) return PA_E_STATE;
  PA_HIP(c, hipSetDevice(c->device));
  const int nb = pa_grid_blocks(c->G.ncell);
  int rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_PART2], &c->cap[SCR_PART2], (size_t)3 * PA_MAX_GRID * sizeof(double)))) return rc;
  double* part = (double*)c->scr[SCR_PART2];
  if (nb + 1 > 3 * PA_MAX_GRID) { pa_set_err(c, "pa_vec_dot: grid too large for the partials buffer"); return PA_E_STATE; }
  if (c->dtype == PA_F64)
    hipLaunchKernelGGL(k_vec_dot<double>, dim3(nb), dim3(PA_BLOCK), 0, c->stream, (const double*)a, (const double*)b, diff,
                       c->G.ncell, part);
  else
    hipLaunchKernelGGL(k_vec_dot<float>, dim3(nb), dim3(PA_BLOCK), 0, c->stream, (const float*)a, (const float*)b, diff,
                       c->G.ncell, part);
  hipLaunchKernelGGL(k_vec_dot_final, dim3(1), dim3(PA_BLOCK), 0, c->stream, (const double*)part, nb, part + nb);
  PA_HIP(c, hipMemcpyAsync(result, part + nb, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  PA_HIP(c, hipStreamSynchronize(c->stream));
  return PA_OK;
}

int pa_aop(pa_ctx* c, const void* x, void* y, int interior_only) {
  if (!c || !c->grid_set || !c->eq_set) { if (c) pa_set_err(c, "pa_aop: grid/equation not set"); return PA_E_STATE; }
  if (int rc0 = pa_check_eq_applicable(c)) return rc0;
  PA_HIP(c, hipSetDevice(c->device));
  return c->dtype == PA_F64 ? aop_t<double>(c, (const double*)x, (double*)y, interior_only, c->nterms, c->terms)
                            : aop_t<float>(c, (const float*)x, (float*)y, interior_only, c->nterms, c->terms);
}

int pa_rhs_adjust(pa_ctx* c, void* rhs) {
  if (!c || !c->grid_set || !c->eq_set) { if (c) pa_set_err(c, "pa_rhs_adjust: grid/equation not set"); return PA_E_STATE; }
  PA_HIP(c, hipSetDevice(c->device));
  return c->dtype == PA_F64 ? rhs_adjust_t<double>(c, (double*)rhs) : rhs_adjust_t<float>(c, (float*)rhs);
}

int pa_laplacian(pa_ctx* c, const void* x, void* y, int edge) {
  if (!c || !c->grid_set) return PA_E_STATE;
  PA_HIP(c, hipSetDevice(c->device));
  return c->dtype == PA_F64 ? lap_t<double>(c, (const double*)x, (double*)y, edge)
                            : lap_t<float>(c, (const float*)x, (float*)y, edge);
}

int pa_grad(pa_ctx* c, const void* x, void* y, int edge) {
  if (!c || !c->grid_set) return PA_E_STATE;
  PA_HIP(c, hipSetDevice(c->device));
  return c->dtype == PA_F64 ? grad_t<double>(c, (const double*)x, (double*)y, edge)
                            : grad_t<float>(c, (const float*)x, (float*)y, edge);
}

int pa_div(pa_ctx* c, int kind, double u, const void* u_field, const void* x, void* y) {
  if (!c || !c->grid_set) return PA_E_STATE;
  int rc = check_div_kind(c, kind);
  if (rc) return rc;
  PA_HIP(c, hipSetDevice(c->device));
  pa_term t;
  memset(&t, 0, sizeof(t));
  t.kind = kind; t.sign = 1.0; t.u = u; t.u_field = u_field;
  return c->dtype == PA_F64 ? aop_t<double>(c, (const double*)x, (double*)y, 0, 1, &t)
                            : aop_t<float>(c, (const float*)x, (float*)y, 0, 1, &t);
}

int pa_div_edge(pa_ctx* c, double u, const void* u_field, const void* x, void* y) {
  if (!c || !c->grid_set) return PA_E_STATE;
  if (c->ndim != 1) {
    pa_set_err(c, "edge=True Div of a scalar field is 1-D only (the reference raises IndexError, fdc.py:296-303)");
    return PA_E_ARG;
  }
  if (c->G.n2 < 3) { pa_set_err(c, "edge Div needs >= 3 nodes"); return PA_E_ARG; }
  PA_HIP(c, hipSetDevice(c->device));
  pa_term t;
  memset(&t, 0, sizeof(t));
  t.kind = PA_OP_GRAD; t.sign = 1.0;
  if (c->dtype == PA_F64) {
    DevEq<double> E;
    pa_build_eq<double>(c, 1, &t, E);
    hipLaunchKernelGGL(k_edge<double>, dim3(pa_grid_blocks(c->G.ncell)), dim3(PA_BLOCK), 0, c->stream, c->G, E,
                       (const double*)x, (double*)y, c->ndim, 2, (double)u, (const double*)u_field);
  } else {
    DevEq<float> E;
    pa_build_eq<float>(c, 1, &t, E);
    hipLaunchKernelGGL(k_edge<float>, dim3(pa_grid_blocks(c->G.ncell)), dim3(PA_BLOCK), 0, c->stream, c->G, E,
                       (const float*)x, (float*)y, c->ndim, 2, (float)u, (const float*)u_field);
  }
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

int pa_euler_step(pa_ctx* c, const void* in, void* out, int kind, double u, const void* u_field, double nu,
                  double dt) {
  if (!c || !c->grid_set) return PA_E_STATE;
  int rc = check_div_kind(c, kind);
  if (rc) return rc;
  if (in == out) { pa_set_err(c, "pa_euler_step: in-place step is not allowed"); return PA_E_ARG; }
  PA_HIP(c, hipSetDevice(c->device));
  return c->dtype == PA_F64 ? euler_t<double>(c, (const double*)in, (double*)out, kind, u, u_field, nu, dt)
                            : euler_t<float>(c, (const float*)in, (float*)out, kind, u, u_field, nu, dt);
}

int pa_euler_march(pa_ctx* c, void* phi, void* tmp, int kind, double u, const void* u_field, double nu, double dt,
                   int64_t nsteps) {
  if (!c || !c->grid_set) return PA_E_STATE;
  int rc = check_div_kind(c, kind);
  if (rc) return rc;
  if (phi == tmp || nsteps < 0) { pa_set_err(c, "pa_euler_march: bad buffers / step count"); return PA_E_ARG; }
  PaRange range_("pyapes explicit Euler march");
  PA_HIP(c, hipSetDevice(c->device));
  void* buf[2] = {phi, tmp};
  // "BC on load" (pa_sf_kernel.h): when every face has a scalar dirichlet / neumann / symmetry BC the steps of a
  // march need no fill between them -- each forms the face values it reads from its own operands, bit for bit what
  // the fill would have stored -- and ONE ordered fill after the last step completes the result.
  bool bcl = c->bcl && c->sf && !c->slab && c->ndim == 3 && kind == PA_OP_DIV_UPWIND && nsteps >= 2 &&
             c->G.n0 >= 5 && c->G.n1 >= 5 && c->G.n2 >= 5;
  for (int f = 0; f < 6 && bcl; ++f)
    bcl = c->bc[f].type >= PA_BC_DIRICHLET && c->bc[f].type <= PA_BC_SYMMETRY && !c->bc[f].vals;
  for (int64_t s = 0; s < nsteps; ++s) {
    if (bcl) {
      const int fr = c->dtype == PA_F64
                         ? euler_bcl_t<double>(c, (const double*)buf[s & 1], (double*)buf[(s + 1) & 1], kind, u, u_field, nu, dt)
                         : euler_bcl_t<float>(c, (const float*)buf[s & 1], (float*)buf[(s + 1) & 1], kind, u, u_field, nu, dt);
      if (fr < 0) return fr;
      if (fr > 0) continue;
      if (s > 0) { pa_set_err(c, "pa_euler_march: the BC-on-load step declined in the middle of a march"); return PA_E_STATE; }
      bcl = false;   // not for k_sf (row length, alignment ...): the classic sequence from the first step on
    }
    rc = c->dtype == PA_F64
             ? euler_t<double>(c, (const double*)buf[s & 1], (double*)buf[(s + 1) & 1], kind, u, u_field, nu, dt)
             : euler_t<float>(c, (const float*)buf[s & 1], (float*)buf[(s + 1) & 1], kind, u, u_field, nu, dt);
    if (rc) return rc;
  }
  if (bcl && nsteps > 0) {
    PA_HIP(c, hipGetLastError());
    return c->dtype == PA_F64 ? pa_bc_apply_auto<double>(c, (double*)buf[nsteps & 1], false)
                              : pa_bc_apply_auto<float>(c, (float*)buf[nsteps & 1], false);
  }
  return PA_OK;
}

}  // extern "C"
