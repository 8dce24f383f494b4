// pa_core.hip -- libpyapes_hip: generic (any dimension / extent / BC mix) kernels and
// the host side of the C ABI declared in include/pyapes_hip.h.  gfx950 only.
//
// Reference seams (paths relative to the reference repo) are cited per function.
// The 3-D fast-path kernels for the CG phases live in pa_cg3d.hip.
#include "pa_host.h"
#include "pa_epilogue.h"

#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>

// ============================================================================
//  kernels
// ============================================================================

// ---- BC fill of one face (pyapes/variables/bcs.py:200-280) --------------------------
template <typename T>
struct BCArgs {
  int axis, side, type;
  T sval;            // dirichlet value, or neumann additive constant for scalar V
  const T* vals;     // per-node values (dirichlet g / neumann V) or null
  T c23, dxf, ndir;  // neumann with per-node V: ((2/3)*V)*dxf*ndir
  T c43, c13;
  const T* far0;     // periodic axis-0 on a slab: planes that live on the other end rank
  const T* far1;
  const int* done;   // when set and *done != 0 the fill is skipped (iterate already final)
};

template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_bc_face(DevGeom G, T* __restrict__ x, BCArgs<T> B) {
  if (B.done && *B.done) return;
  const int a = B.axis;
  const int64_t nu = (a == 0) ? G.n1 : G.n0;
  const int64_t nv = (a == 2) ? G.n1 : G.n2;
  const int64_t N = (a == 0) ? G.g0 : (a == 1 ? G.n1 : G.n2);
  const int64_t off = (a == 0) ? G.off0 : 0;
  const int64_t st = (a == 0) ? G.s0 : (a == 1 ? G.s1 : 1);
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nu * nv;
       q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t u = q / nv, v = q - u * nv;
    int64_t base;  // offset of the node with axis index 0
    if (a == 0) base = u * G.s1 + v;
    else if (a == 1) base = u * G.s0 + v;
    else base = u * G.s0 + v * G.s1;
    const int64_t f = (B.side == 0) ? 0 : N - 1;
    const int64_t p1 = (B.side == 0) ? pa_wrap(1, N) : pa_wrap(N - 2, N);
    const int64_t p2 = (B.side == 0) ? pa_wrap(2 % N, N) : pa_wrap(N - 3 < 0 ? N - 3 + N : N - 3, N);
    T* xf = x + base + (f - off) * st;
    if (B.type == 1) {
      *xf = B.vals ? B.vals[q] : B.sval;
    } else if (B.type == 2) {
      T vp = x[base + (p1 - off) * st];
      T vpp = x[base + (p2 - off) * st];
      T ct;
      if (B.vals) {
        ct = B.c23 * B.vals[q];
        ct = ct * B.dxf;
        ct = ct * B.ndir;
      } else {
        ct = B.sval;
      }
      T t1 = B.c43 * vp;
      T t2 = B.c13 * vpp;
      t1 = t1 - t2;
      *xf = t1 + ct;
    } else if (B.type == 3) {
      *xf = x[base + (p1 - off) * st];
    } else if (B.type == 4) {
      if (B.side == 0) {
        // x[0] = x[1] - x[N-1] + x[N-2]
        T vp = x[base + (p1 - off) * st];
        T vf, vff;
        if (B.far0) {
          vf = B.far0[q];
          vff = B.far1[q];
        } else {
          vf = x[base + (N - 1 - off) * st];
          vff = x[base + (pa_wrap(N - 2, N) - off) * st];
        }
        T t1 = vp - vf;
        *xf = t1 + vff;
      } else if (B.far0) {
        // slab: the lower end rank lives elsewhere; far0 = its x[1], and the new x[0] it computes,
        // x1 - x[N-1] + x[N-2], is recomputed here bit for bit from the planes this rank owns
        T t1 = B.far0[q] - x[base + (N - 1 - off) * st];
        *xf = t1 + x[base + (pa_wrap(N - 2, N) - off) * st];
      } else {
        *xf = x[base + (0 - off) * st];
      }
    }
  }
}

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


// ---- CG: r = (b - A x) on S, d = r, partial sum r.r (linalg.py:98-107) ---------------
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_cg_init(DevGeom G, DevEq<T> E, Vec<T> xv,
                                                       const T* __restrict__ rhs, T* __restrict__ r,
                                                       T* __restrict__ d, T* __restrict__ send_lo,
                                                       T* __restrict__ send_hi, double* __restrict__ partials) {
  FieldAcc<T> acc{xv};
  double s[1] = {0.0};
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < G.ncell;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t i, j, k;
    pa_decode(G, idx, i, j, k);
    T rv = (T)0;
    if (pa_in_S(G, i, j, k)) {
      T ax = pa_apply_terms<T>(G, E, acc, i, j, k, xv.p[idx]);
      rv = rhs[idx] - ax;
      T p = rv * rv;
      s[0] += (double)p;
    }
    r[idx] = rv;
    if (d) d[idx] = rv;
    if (send_lo && i == 0) send_lo[j * G.s1 + k] = rv;
    if (send_hi && i == G.n0 - 1) send_hi[j * G.s1 + k] = rv;
  }
  pa_block_reduce_store<1>(s, partials);
}

// the same, from A x already computed by the tiled kernel (zero outside S) and sitting in `r`: same
// loop, same grid, same partial sums -- r, d and the sum r.r come out bit-identical to k_cg_init
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_cg_init_ax(DevGeom G, const T* __restrict__ rhs, T* __restrict__ r,
                                                          T* __restrict__ d, T* __restrict__ send_lo,
                                                          T* __restrict__ send_hi, double* __restrict__ partials) {
  double s[1] = {0.0};
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < G.ncell;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t i, j, k;
    pa_decode(G, idx, i, j, k);
    T rv = (T)0;
    if (pa_in_S(G, i, j, k)) {
      rv = rhs[idx] - r[idx];
      T p = rv * rv;
      s[0] += (double)p;
    }
    r[idx] = rv;
    if (d) d[idx] = rv;
    if (send_lo && i == 0) send_lo[j * G.s1 + k] = rv;
    if (send_hi && i == G.n0 - 1) send_hi[j * G.s1 + k] = rv;
  }
  pa_block_reduce_store<1>(s, partials);
}

// A kernel argument struct that is indexed with a run-time index (E.t[q], B.f[face]) gets copied to
// scratch memory by the compiler, and a kernel with a private segment costs ~10 us more per dispatch
// on MI355X (measured: 64^3 CG iteration 29 -> 52 us).  Reading the struct in place -- through the
// kernarg segment pointer -- keeps such kernels off scratch.  `off` = byte offset of the parameter.
template <typename S>
__device__ __forceinline__ const S& pa_kernarg(size_t off) {
  return *(const S*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + off);
}
static_assert(sizeof(DevGeom) % 8 == 0, "second kernel parameter starts at sizeof(DevGeom)");

// ---- CG phase A: d' = r + beta d ; partial sum d'.(A d')  (linalg.py:115-120, 141) ----
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_cg_a(DevGeom G, DevEq<T> E, const SolverScalars* __restrict__ sc,
                                                    Vec<T> rv, Vec<T> dv, T* __restrict__ dnew,
                                                    double* __restrict__ partials, CgEpi epi) {
  if (sc->done) return;
  DirAcc<T> acc{rv, dv, (T)sc->beta};
  double s[1] = {0.0};
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < G.ncell;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t i, j, k;
    pa_decode(G, idx, i, j, k);
    T e = (T)0;
    if (pa_in_S(G, i, j, k)) {
      e = acc.at(G, i, j, k);
      T Ad = pa_apply_terms<T>(G, E, acc, i, j, k, e);
      T p = e * Ad;
      s[0] += (double)p;
    }
    dnew[idx] = e;
  }
  pa_block_reduce_store<1>(s, partials);
  pa_cg_epilogue<T>(epi);
}

// ---- CG phase B: x += alpha d ; r -= alpha A d ; partial sums r.r and |dx|^2 off-shell
//      (linalg.py:122-134)
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_cg_b(DevGeom G, DevEq<T> E, const SolverScalars* __restrict__ sc,
                                                    Vec<T> dv, T* __restrict__ x, T* __restrict__ r,
                                                    T* __restrict__ send_lo, T* __restrict__ send_hi,
                                                    double* __restrict__ partials, CgEpi epi) {
  if (sc->done) return;
  FieldAcc<T> acc{dv};
  const T alpha = (T)sc->alpha;
  double s[2] = {0.0, 0.0};
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < G.ncell;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t i, j, k;
    pa_decode(G, idx, i, j, k);
    T rn = (T)0;
    if (pa_in_S(G, i, j, k)) {
      T dc = dv.p[idx];
      T Ad = pa_apply_terms<T>(G, E, acc, i, j, k, dc);
      T xo = x[idx];
      T ad = alpha * dc;
      T xn = xo + ad;
      x[idx] = xn;
      T aAd = alpha * Ad;
      rn = r[idx] - aAd;
      r[idx] = rn;
      T p = rn * rn;
      s[0] += (double)p;
      if (!pa_on_shell(G, i, j, k)) {
        T df = xn - xo;
        T p2 = df * df;
        s[1] += (double)p2;
      }
    }
    if (send_lo && i == 0) send_lo[j * G.s1 + k] = rn;
    if (send_hi && i == G.n0 - 1) send_hi[j * G.s1 + k] = rn;
  }
  pa_block_reduce_store<2>(s, partials);
  pa_cg_epilogue<T>(epi);
}

// ---- boundary shell: sum (x_new - x_old)^2 over shell nodes after the BC fill, and keep
//      x_old for the next iteration (stop test of linalg.py:134 includes boundary nodes) ---
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_shell(DevGeom G, const SolverScalars* __restrict__ sc,
                                                     const T* __restrict__ x, T* __restrict__ shell_old,
                                                     double* __restrict__ partials, int mode) {
  if (mode == 1 && sc->done) return;
  const int64_t sz[3] = {G.n1 * G.n2, G.n0 * G.n2, G.n0 * G.n1};
  int64_t total = 0;
  int64_t start[6];
  for (int f = 0; f < 6; ++f) {
    start[f] = total;
    total += G.act[f >> 1] ? sz[f >> 1] : 0;
  }
  double s[1] = {0.0};
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total;
       q += (int64_t)gridDim.x * blockDim.x) {
    int f = 0;
    for (int w = 1; w < 6; ++w)
      if (G.act[w >> 1] && q >= start[w]) f = w;
    const int a = f >> 1, side = f & 1;
    const int64_t local = q - start[f];
    int64_t i, j, k;
    if (a == 0) {
      // owned by this rank only if it holds the global boundary plane
      int64_t gi = side == 0 ? 0 : G.g0 - 1;
      i = gi - G.off0;
      if (i < 0 || i >= G.n0) continue;
      j = local / G.n2; k = local - j * G.n2;
    } else if (a == 1) {
      i = local / G.n2; k = local - i * G.n2;
      j = side == 0 ? 0 : G.n1 - 1;
      int64_t gi = i + G.off0;
      if (G.act[0] && (gi == 0 || gi == G.g0 - 1)) continue;  // owned by an axis-0 face
    } else {
      i = local / G.n1; j = local - i * G.n1;
      k = side == 0 ? 0 : G.n2 - 1;
      int64_t gi = i + G.off0;
      if (G.act[0] && (gi == 0 || gi == G.g0 - 1)) continue;
      if (G.act[1] && (j == 0 || j == G.n1 - 1)) continue;
    }
    // a node on both the lower and the upper face of one axis (extent 1) cannot occur: act => n > 1
    const int64_t o = i * G.s0 + j * G.s1 + k;
    T xn = x[o];
    if (mode == 1) {
      T df = xn - shell_old[q];
      T p = df * df;
      s[0] += (double)p;
    }
    shell_old[q] = xn;
  }
  if (mode == 1) pa_block_reduce_store<1>(s, partials);
}


// ---- fused BC fill + boundary-shell stop-test term ---------------------------------------------
// The reference applies the faces one after the other (linalg.py:295-297); a later face reads, on
// the shared edges, what an earlier face wrote.  For the factory order xl,xu,yl,yu,zl,zu the final
// value of a shell node is a closed form of ORIGINAL interior values: v3 = zfill(v2), v2 =
// yfill(v1), v1 = xfill(v0).  k_bc_compute evaluates that per shell node from the unmodified
// field into a compact shell buffer (and accumulates (new - old)^2 against the previous
// iteration's shell for the stop test); k_bc_scatter writes the shell back.  2 launches instead
// of 6 face fills + 1 shell pass, same values bit for bit.
template <typename T>
struct BCFaceDev {
  int type;
  T sval;          // dirichlet value / neumann additive constant for scalar V
  const T* vals;   // per-node g or V
  T dxf, ndir;
};
template <typename T>
struct BCAll {
  BCFaceDev<T> f[6];
  T c43, c13, c23;
  const T* far_lo0;  // slab, periodic axis 0: x[N-1], x[N-2] (lower end rank), x[1] (upper end rank)
  const T* far_lo1;
  const T* far_hi0;
  int slab_periodic0;
};

template <typename T>
__device__ __forceinline__ T pa_bc_const(const BCAll<T>& B, int f, int64_t q) {
  const BCFaceDev<T>& F = B.f[f];
  if (!F.vals) return F.sval;
  T ct = B.c23 * F.vals[q];
  ct = ct * F.dxf;
  return ct * F.ndir;
}

// Closed-form evaluation, one stage per axis.  stage<A>(i,j,k) = value of the node after the
// faces of axes 0..A have been applied, expressed through stage<A-1> of the nodes that face reads.
template <typename T>
struct BCEval {
  const DevGeom& G;
  const BCAll<T>& B;
  const T* __restrict__ x;

  // axis-0 neighbour of (., j, k) at GLOBAL plane g before any fill; on a slab whose ring is cut the
  // planes of the other end rank come from the exchanged far buffers
  __device__ __forceinline__ T raw0(int f, int64_t g, int64_t base) const {
    // one pointer select, one load (an if/return ladder here was miscompiled by hipcc 7.2 when fully
    // inlined: tests/test_gpu_bc_fused.py is the regression test)
    const T* p = x + (g - G.off0) * G.s0;
    if (B.slab_periodic0 && B.f[f].type == 4) {
      if (f == 0) {
        p = (g == G.g0 - 1) ? B.far_lo0 : ((g == G.g0 - 2) ? B.far_lo1 : p);
      } else {
        p = (g == 1) ? B.far_hi0 : p;
      }
    }
    return p[base];
  }

  // the value face f writes, from the three pre-axis values it can read:
  //   p1 = prev, p2 = prev2 (neumann / symmetry); periodic: a = x[1], b = x[N-1], c = x[N-2]
  __device__ __forceinline__ T stage0(int64_t i, int64_t j, int64_t k) const {
    const int64_t gi = i + G.off0;
    const int64_t base = j * G.s1 + k;
    int f = -1;
    if (G.act[0]) {
      if (gi == 0 && B.f[0].type) f = 0;
      else if (gi == G.g0 - 1 && B.f[1].type) f = 1;
    }
    if (f < 0) return x[i * G.s0 + base];
    const int type = B.f[f].type;
    const bool lower = f == 0;
    const int64_t N = G.g0;
    if (type == 1) return B.f[f].vals ? B.f[f].vals[base] : B.f[f].sval;
    if (type == 2) {
      T t1 = B.c43 * raw0(f, lower ? 1 : N - 2, base);
      T t2 = B.c13 * raw0(f, lower ? 2 : N - 3, base);
      t1 = t1 - t2;
      return t1 + pa_bc_const<T>(B, f, base);
    }
    if (type == 3) return raw0(f, lower ? 1 : N - 2, base);
    T t1 = raw0(f, 1, base) - raw0(f, N - 1, base);
    return t1 + raw0(f, N - 2, base);
  }

  __device__ __forceinline__ T stage1(int64_t i, int64_t j, int64_t k) const {
    int f = -1;
    if (G.act[1]) {
      if (j == 0 && B.f[2].type) f = 2;
      else if (j == G.n1 - 1 && B.f[3].type) f = 3;
    }
    if (f < 0) return stage0(i, j, k);
    const int type = B.f[f].type;
    const bool lower = f == 2;
    const int64_t N = G.n1, q = i * G.n2 + k;
    if (type == 1) return B.f[f].vals ? B.f[f].vals[q] : B.f[f].sval;
    if (type == 2) {
      T t1 = B.c43 * stage0(i, lower ? 1 : N - 2, k);
      T t2 = B.c13 * stage0(i, lower ? 2 : N - 3, k);
      t1 = t1 - t2;
      return t1 + pa_bc_const<T>(B, f, q);
    }
    if (type == 3) return stage0(i, lower ? 1 : N - 2, k);
    T t1 = stage0(i, 1, k) - stage0(i, N - 1, k);
    return t1 + stage0(i, N - 2, k);
  }

  __device__ __forceinline__ T stage2(int64_t i, int64_t j, int64_t k) const {
    int f = -1;
    if (G.act[2]) {
      if (k == 0 && B.f[4].type) f = 4;
      else if (k == G.n2 - 1 && B.f[5].type) f = 5;
    }
    if (f < 0) return stage1(i, j, k);
    const int type = B.f[f].type;
    const bool lower = f == 4;
    const int64_t N = G.n2, q = i * G.n1 + j;
    if (type == 1) return B.f[f].vals ? B.f[f].vals[q] : B.f[f].sval;
    if (type == 2) {
      T t1 = B.c43 * stage1(i, j, lower ? 1 : N - 2);
      T t2 = B.c13 * stage1(i, j, lower ? 2 : N - 3);
      t1 = t1 - t2;
      return t1 + pa_bc_const<T>(B, f, q);
    }
    if (type == 3) return stage1(i, j, lower ? 1 : N - 2);
    T t1 = stage1(i, j, 1) - stage1(i, j, N - 1);
    return t1 + stage1(i, j, N - 2);
  }
};

template <typename T>
__device__ __forceinline__ T pa_bc_v3(const DevGeom& G, const BCAll<T>& B, const T* __restrict__ x, int64_t i,
                                      int64_t j, int64_t k) {
  BCEval<T> ev{G, B, x};
  return ev.stage2(i, j, k);
}

// enumerate the shell nodes this rank owns exactly once (same layout as k_shell)
__device__ __forceinline__ bool pa_shell_node(const DevGeom& G, int64_t q, const int64_t* start, int64_t& i,
                                              int64_t& j, int64_t& k) {
  int f = 0;
  for (int w = 1; w < 6; ++w)
    if (G.act[w >> 1] && q >= start[w]) f = w;
  const int a = f >> 1, side = f & 1;
  const int64_t local = q - start[f];
  if (a == 0) {
    int64_t gi = side == 0 ? 0 : G.g0 - 1;
    i = gi - G.off0;
    if (i < 0 || i >= G.n0) return false;
    j = local / G.n2; k = local - j * G.n2;
  } else if (a == 1) {
    i = local / G.n2; k = local - i * G.n2;
    j = side == 0 ? 0 : G.n1 - 1;
    int64_t gi = i + G.off0;
    if (G.act[0] && (gi == 0 || gi == G.g0 - 1)) return false;
  } else {
    i = local / G.n1; j = local - i * G.n1;
    k = side == 0 ? 0 : G.n2 - 1;
    int64_t gi = i + G.off0;
    if (G.act[0] && (gi == 0 || gi == G.g0 - 1)) return false;
    if (G.act[1] && (j == 0 || j == G.n1 - 1)) return false;
  }
  return true;
}

template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_bc_compute(DevGeom G, BCAll<T> B_, const int* __restrict__ done,
                                                          const T* x, const T* __restrict__ shell_old,
                                                          T* __restrict__ shell_new, double* __restrict__ partials,
                                                          int with_delta, T* xw) {
  if (done && *done) return;
  const BCAll<T>& B = pa_kernarg<BCAll<T>>(sizeof(DevGeom));
  const int64_t sz[3] = {G.n1 * G.n2, G.n0 * G.n2, G.n0 * G.n1};
  int64_t total = 0, start[6];
  for (int f = 0; f < 6; ++f) { start[f] = total; total += G.act[f >> 1] ? sz[f >> 1] : 0; }
  double s[1] = {0.0};
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total;
       q += (int64_t)gridDim.x * blockDim.x) {
    int64_t i, j, k;
    if (!pa_shell_node(G, q, start, i, j, k)) continue;
    T v = pa_bc_v3<T>(G, B, x, i, j, k);
#ifdef PA_DEBUG_BC
    if (i == 0 && j == 1 && k == G.n2 - 1) {
      BCEval<T> ev{G, B, x};
      printf("DBG node(0,1,%lld) v=%g stage1(0,1,n2-2)=%g stage0=%g types %d %d %d %d %d %d slabp %d far %p %p %p raw1 %g rawN1 %g rawN2 %g\n",
             (long long)k, (double)v, (double)ev.stage1(0, 1, G.n2 - 2), (double)ev.stage0(0, 1, G.n2 - 2),
             B.f[0].type, B.f[1].type, B.f[2].type, B.f[3].type, B.f[4].type, B.f[5].type, B.slab_periodic0,
             (void*)B.far_lo0, (void*)B.far_lo1, (void*)B.far_hi0, (double)ev.raw0(0, 1, 1 * G.s1 + G.n2 - 2),
             (double)ev.raw0(0, G.g0 - 1, 1 * G.s1 + G.n2 - 2), (double)ev.raw0(0, G.g0 - 2, 1 * G.s1 + G.n2 - 2));
    }
#endif
    shell_new[q] = v;
    if (xw) xw[i * G.s0 + j * G.s1 + k] = v;  // single pass (see bc_shell_fused): no later read sees this node
    if (with_delta) {
      T df = v - shell_old[q];
      T p = df * df;
      s[0] += (double)p;
    }
  }
  if (with_delta) pa_block_reduce_store<1>(s, partials);
}

template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_bc_scatter(DevGeom G, const int* __restrict__ done,
                                                          T* __restrict__ x, const T* __restrict__ shell_new) {
  if (done && *done) return;
  const int64_t sz[3] = {G.n1 * G.n2, G.n0 * G.n2, G.n0 * G.n1};
  int64_t total = 0, start[6];
  for (int f = 0; f < 6; ++f) { start[f] = total; total += G.act[f >> 1] ? sz[f >> 1] : 0; }
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total;
       q += (int64_t)gridDim.x * blockDim.x) {
    int64_t i, j, k;
    if (!pa_shell_node(G, q, start, i, j, k)) continue;
    x[i * G.s0 + j * G.s1 + k] = shell_new[q];
  }
}


// ---- BC fill, one launch per AXIS (lower then upper face in the same thread) + the boundary-shell
//      part of the stop test in the same pass ------------------------------------------------------
// Valid for the factory order xl,xu,yl,yu,zl,zu: within an axis the upper fill only depends on the
// lower fill through the periodic copy x[N-1] = x[0], which the same thread has just computed;
// across axes the launches are still sequential.  Halves the strided passes over the z faces and
// removes the separate k_shell pass: every shell node is charged to the LAST face that writes it
// (z over y over x), which is where its final value is known.
template <typename T>
struct BCPairArgs {
  BCArgs<T> lo, hi;   // .type == 0: face absent on this rank
  int axis;
  int64_t pos_lo, pos_hi;  // offsets of the two faces in the shell buffer
};

template <typename T>
__device__ __forceinline__ T pa_bc_face_value(const BCArgs<T>& B, const T* __restrict__ x, int64_t base, int64_t st,
                                              int64_t off, int64_t N, int64_t q, T x0_new) {
  const bool lower = B.side == 0;
  const int64_t p1 = lower ? 1 : N - 2, p2 = lower ? 2 : N - 3;
  if (B.type == 1) return B.vals ? B.vals[q] : B.sval;
  if (B.type == 2) {
    T ct;
    if (B.vals) {
      ct = B.c23 * B.vals[q];
      ct = ct * B.dxf;
      ct = ct * B.ndir;
    } else {
      ct = B.sval;
    }
    T t1 = B.c43 * x[base + (p1 - off) * st];
    T t2 = B.c13 * x[base + (p2 - off) * st];
    t1 = t1 - t2;
    return t1 + ct;
  }
  if (B.type == 3) return x[base + (p1 - off) * st];
  // periodic
  if (lower) {
    T vp = x[base + (1 - off) * st];
    T vf = B.far0 ? B.far0[q] : x[base + (N - 1 - off) * st];
    T vff = B.far0 ? B.far1[q] : x[base + (N - 2 - off) * st];
    T t1 = vp - vf;
    return t1 + vff;
  }
  if (B.far0) {  // slab: x[1] of the lower end rank arrived in far0; recompute its new x[0] bit for bit
    T t1 = B.far0[q] - x[base + (N - 1 - off) * st];
    return t1 + x[base + (N - 2 - off) * st];
  }
  return x0_new;
}

template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_bc_pair(DevGeom G, T* __restrict__ x, BCPairArgs<T> P,
                                                       const int* __restrict__ done, T* __restrict__ shell_old,
                                                       double* __restrict__ partials, int mode, CgEpi epi) {
  if (done && *done) return;
  const int a = P.axis;
  const int64_t nu = (a == 0) ? G.n1 : G.n0;
  const int64_t nv = (a == 2) ? G.n1 : G.n2;
  const int64_t N = (a == 0) ? G.g0 : (a == 1 ? G.n1 : G.n2);
  const int64_t off = (a == 0) ? G.off0 : 0;
  const int64_t st = (a == 0) ? G.s0 : (a == 1 ? G.s1 : 1);
  double s[1] = {0.0};
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nu * nv;
       q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t u = q / nv, v = q - u * nv;
    int64_t base;
    bool owned = true;  // is this axis the last one whose faces contain the node?
    if (a == 0) {
      base = u * G.s1 + v;
      if (G.act[1] && (u == 0 || u == G.n1 - 1)) owned = false;
      if (G.act[2] && (v == 0 || v == G.n2 - 1)) owned = false;
    } else if (a == 1) {
      base = u * G.s0 + v;
      if (G.act[2] && (v == 0 || v == G.n2 - 1)) owned = false;
    } else {
      base = u * G.s0 + v * G.s1;
    }
    T x0_new = (T)0;
    if (P.lo.type) {
      T* xf = x + base + (0 - off) * st;
      T val;
      if (mode == 3) {
        val = *xf;
      } else {
        val = pa_bc_face_value<T>(P.lo, x, base, st, off, N, q, (T)0);
        *xf = val;
      }
      x0_new = val;
      if (mode && owned) {
        if (mode == 1) {
          T df = val - shell_old[P.pos_lo + q];
          T p = df * df;
          s[0] += (double)p;
        }
        shell_old[P.pos_lo + q] = val;
      }
    } else if (P.hi.type == 4 && !P.hi.far0) {
      x0_new = x[base + (0 - off) * st];
    }
    if (P.hi.type) {
      T* xf = x + base + (N - 1 - off) * st;
      T val;
      if (mode == 3) {
        val = *xf;
      } else {
        val = pa_bc_face_value<T>(P.hi, x, base, st, off, N, q, x0_new);
        *xf = val;
      }
      if (mode && owned) {
        if (mode == 1) {
          T df = val - shell_old[P.pos_hi + q];
          T p = df * df;
          s[0] += (double)p;
        }
        shell_old[P.pos_hi + q] = val;
      }
    }
  }
  if (mode == 1) {
    pa_block_reduce_store<1>(s, partials);
    pa_cg_epilogue<T>(epi);
  }
}

// ---- axisymmetric meshes: the r-dependent coefficient rows, once per mesh -------------------------
// Row q of the 6 x n_r table (literal expressions of the reference, evaluated in the grid dtype):
//   0  (1 + s) / dr^2, s = nan_to_num(dr / (2 r))       Laplacian Ap along r   (tools.py:86-99)
//   1  (1 - s) / dr^2                                    Laplacian Am along r   (tools.py:101-106)
//   2  (2/3 + nan_to_num(2/3 dr / r)) / dr^2            neumann / symmetry row (fdc.py:395-417)
//   3  2/3 - nan_to_num(1/3 dr / r)                      rhs adjustment factor  (fdc.py:440-453)
//   4  nan_to_num(2 dr / r)                              Ac of Div along r      (tools.py:64-78)
//   5  r
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_rz_tables(int64_t nr, const T* __restrict__ r_nodes, T dr,
                                                         T* __restrict__ tab) {
  auto nn = [](T v) -> T { return (isnan(v) || isinf(v)) ? (T)0 : v; };
  const T h2 = dr * dr;                       // dx[0] ** 2
  const T c23 = (T)(2.0 / 3.0), c13 = (T)(1.0 / 3.0);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nr; i += (int64_t)gridDim.x * blockDim.x) {
    const T r = r_nodes[i];
    T t = (T)2 * r;
    const T s = nn(dr / t);
    T ap = (T)1 + s;
    T am = (T)1 - s;
    tab[i] = ap / h2;
    tab[nr + i] = am / h2;
    t = c23 * dr;
    T a = nn(t / r);
    a = c23 + a;
    tab[2 * nr + i] = a / h2;
    t = c13 * dr;
    a = nn(t / r);
    tab[3 * nr + i] = c23 - a;
    t = (T)2 * dr;
    tab[4 * nr + i] = nn(t / r);
    tab[5 * nr + i] = r;
  }
}

// ---- reductions of per-block partials + scalar logic ------------------------------------
// sums[slot[s]] (+)= sum over blocks of partials[b*ns + s]
__device__ __forceinline__ double pa_reduce_partials(const double* __restrict__ partials, int nblk, int ns,
                                                     int s, double* sm) {
  double v = 0.0;
  for (int b = threadIdx.x; b < nblk; b += blockDim.x) v += partials[(int64_t)b * ns + s];
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sm[w];
  __syncthreads();
  return t;  // valid on thread 0
}

template <typename T>
__device__ __forceinline__ double pa_nan_to_num(T v) {
  return (isnan(v) || isinf(v)) ? 0.0 : (double)v;  // linalg.py:302-305
}

// stage 0: reduce only (multi-GPU, before the all-reduce); 1: logic only; 2: both
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_cg_post_a(SolverScalars* sc, const double* partials, int nblk,
                                                         double* sums, int stage) {
  __shared__ double sm[PA_BLOCK / 64];
  if (sc->done) return;
  if (stage != 1) {
    double v = pa_reduce_partials(partials, nblk, 1, 0, sm);
    if (threadIdx.x == 0) sums[0] = v;
  }
  if (stage != 0 && threadIdx.x == 0) pa_logic_a<T>(sc, sums);
}

template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_cg_post_b(SolverScalars* sc, const double* partials, int nblk,
                                                         const double* partials_shell, int nblk_shell,
                                                         double* sums, int stage) {
  __shared__ double sm[PA_BLOCK / 64];
  if (sc->done) return;
  if (stage != 1) {
    double rr = pa_reduce_partials(partials, nblk, 2, 0, sm);
    double dx2 = pa_reduce_partials(partials, nblk, 2, 1, sm);
    double sh = nblk_shell > 0 ? pa_reduce_partials(partials_shell, nblk_shell, 1, 0, sm) : 0.0;
    if (threadIdx.x == 0) {
      sums[1] = rr;
      sums[2] = dx2 + sh;
    }
  }
  if (stage != 0 && threadIdx.x == 0) pa_logic_b<T>(sc, sums);
}

// overlap mode: the part of post_b the next phase A needs (sum r.r -> beta) ...
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_cg_post_b1(SolverScalars* sc, const double* partials, int nblk,
                                                          double* sums) {
  __shared__ double sm[PA_BLOCK / 64];
  if (sc->done) return;
  double rr = pa_reduce_partials(partials, nblk, 2, 0, sm);
  double dx2 = pa_reduce_partials(partials, nblk, 2, 1, sm);
  if (threadIdx.x == 0) {
    sums[1] = rr;
    sums[2] = dx2;
    T rr_new = (T)rr;
    T rr_old = (T)sc->rr;
    sc->rr_old = (double)rr_old;
    sc->beta = (double)(rr_new / rr_old);
    sc->rr = (double)rr_new;
  }
}

// ... and the part that needs the BC-filled boundary shell (stop test, iteration count, done)
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_cg_post_b2(SolverScalars* sc, const double* partials_shell,
                                                          int nblk_shell, const double* sums) {
  __shared__ double sm[PA_BLOCK / 64];
  if (sc->done) return;
  double sh = nblk_shell > 0 ? pa_reduce_partials(partials_shell, nblk_shell, 1, 0, sm) : 0.0;
  if (threadIdx.x == 0) {
    T tol = (T)sqrt(sums[2] + sh);
    sc->tol = (double)tol;
    if (isnan(tol) || isinf(tol)) {
      sc->err = 1;
      sc->done = 1;
      return;
    }
    sc->itr += 1;
    if (sc->itr > sc->max_it || !(sc->tol > sc->tolerance)) sc->done = 1;
  }
}

template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_cg_post_init(SolverScalars* sc, const double* partials, int nblk,
                                                            double* sums, int stage) {
  __shared__ double sm[PA_BLOCK / 64];
  if (stage != 1) {
    double v = pa_reduce_partials(partials, nblk, 1, 0, sm);
    if (threadIdx.x == 0) sums[1] = v;
  }
  if (stage != 0 && threadIdx.x == 0) sc->rr = (double)(T)sums[1];
}

// ---- Jacobi sweep [new, SURVEY a15] -----------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_jacobi(DevGeom G, DevEq<T> E, const SolverScalars* __restrict__ sc,
                                                      Vec<T> xv, const T* __restrict__ rhs,
                                                      T* __restrict__ xnew, T omega,
                                                      double* __restrict__ partials) {
  if (sc->done) return;
  FieldAcc<T> acc{xv};
  double s[2] = {0.0, 0.0};
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < G.ncell;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t i, j, k;
    pa_decode(G, idx, i, j, k);
    T xo = xv.p[idx];
    T xn = xo;
    if (pa_in_S(G, i, j, k)) {
      int64_t g[3], N[3];
      pa_gidx(G, i, j, k, g, N);
      T diag = (T)0;
      for (int q = 0; q < E.nterms; ++q) {
        const DevTerm<T>& t = E.t[q];
        T dg = (T)0;
        for (int a = 0; a < 3; ++a) {
          if (!G.act[a]) continue;
          int rc = pa_row_case(G, a, g[a], N[a], G.treat);
          T cB = (E.rz && a == PA_RZ_AXIS) ? E.rz[2 * E.rz_n + g[a]] : E.lap.c23[a];
          T cC = rc == 0 ? E.lap.m2inv[a] : -cB;
          dg = dg + cC;
        }
        if (t.has_coeff) dg = dg * (t.coeff_f ? t.coeff_f[idx] : t.coeff);
        dg = dg * t.sign;
        diag = diag + dg;
      }
      T ax = pa_apply_terms<T>(G, E, acc, i, j, k, xo);
      T res = rhs[idx] - ax;
      res = res / diag;
      T w = omega * res;
      xn = xo + w;
      if (!pa_on_shell(G, i, j, k)) {
        T df = xn - xo;
        T p2 = df * df;
        s[1] += (double)p2;
      }
    }
    xnew[idx] = xn;
  }
  pa_block_reduce_store<2>(s, partials);
}

template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_jacobi_post(SolverScalars* sc, const double* partials, int nblk,
                                                           const double* partials_shell, int nblk_shell,
                                                           double* sums) {
  __shared__ double sm[PA_BLOCK / 64];
  if (sc->done) return;
  double dx2 = pa_reduce_partials(partials, nblk, 2, 1, sm);
  double sh = nblk_shell > 0 ? pa_reduce_partials(partials_shell, nblk_shell, 1, 0, sm) : 0.0;
  if (threadIdx.x == 0) {
    sums[2] = dx2 + sh;
    T tol = (T)sqrt(sums[2]);
    sc->tol = (double)tol;
    if (isnan(tol) || isinf(tol)) { sc->err = 1; sc->done = 1; return; }
    sc->itr += 1;
    if (sc->itr > sc->max_it || !(sc->tol > sc->tolerance)) sc->done = 1;
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

// slab: ghost planes of the new direction, d'_g = r_g + beta d_g -- bitwise what the neighbour
// rank computes for its own boundary plane, so no direction planes are ever exchanged
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_ghost_dir(const SolverScalars* __restrict__ sc, int64_t n,
                                                         const T* __restrict__ r_lo, const T* __restrict__ r_hi,
                                                         const T* __restrict__ d_lo, const T* __restrict__ d_hi,
                                                         T* __restrict__ o_lo, T* __restrict__ o_hi) {
  if (sc->done) return;
  const T beta = (T)sc->beta;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
    if (r_lo) { T b = beta * d_lo[q]; o_lo[q] = r_lo[q] + b; }
    if (r_hi) { T b = beta * d_hi[q]; o_hi[q] = r_hi[q] + b; }
  }
}

// slab, periodic axis 0: copies of the x planes the other end rank's BC fill needs, placed next to
// the residual planes in the packed send buffers (one message per neighbour and iteration)
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_pack_planes(const SolverScalars* __restrict__ sc, int64_t n,
                                                           const T* __restrict__ s0, T* __restrict__ d0,
                                                           const T* __restrict__ s1, T* __restrict__ d1,
                                                           const T* __restrict__ s2, T* __restrict__ d2) {
  if (sc->done) return;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
    if (d0) d0[q] = s0[q];
    if (d1) d1[q] = s1[q];
    if (d2) d2[q] = s2[q];
  }
}

template <typename T>
__global__ void k_copy(const T* __restrict__ a, T* __restrict__ b, int64_t n) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
       idx += (int64_t)gridDim.x * blockDim.x)
    b[idx] = a[idx];
}

// ---- BiCGSTAB kernels (linalg.py:162-279) ------------------------------------------------
// p' = r + beta (p - omega v) (with neighbours, so A p' needs no second pass); v' = A p' on S;
// partial sum r0.v'
template <typename T>
struct BicgPAcc {
  Vec<T> r, p, v;
  T beta, omega;
  __device__ __forceinline__ T at(const DevGeom& G, int64_t i, int64_t j, int64_t k) const {
    const int64_t o = j * G.s1 + k;  // pointers first, one load per field after (see DirAcc)
    const T* rb = r.p + i * G.s0;
    const T* pb = p.p + i * G.s0;
    const T* vb = v.p + i * G.s0;
    if (i < 0) { rb = r.glo; pb = p.glo; vb = v.glo; }
    if (i >= G.n0) { rb = r.ghi; pb = p.ghi; vb = v.ghi; }
    const T rv = rb[o], pv = pb[o], vv = vb[o];
    T t = omega * vv;
    t = pv - t;
    t = beta * t;
    return rv + t;
  }
};

template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_bicg_pv(DevGeom G, DevEq<T> E, const SolverScalars* __restrict__ sc,
                                                       Vec<T> rv, Vec<T> pv, Vec<T> vv, const T* __restrict__ r0,
                                                       T* __restrict__ pnew, T* __restrict__ vnew,
                                                       double* __restrict__ partials) {
  if (sc->done) return;
  BicgPAcc<T> acc{rv, pv, vv, (T)sc->beta, (T)sc->omega};
  double s[1] = {0.0};
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < G.ncell;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t i, j, k;
    pa_decode(G, idx, i, j, k);
    T pc = acc.at(G, i, j, k);
    T vn = (T)0;
    if (pa_in_S(G, i, j, k)) {
      vn = pa_apply_terms<T>(G, E, acc, i, j, k, pc);
      T p = r0[idx] * vn;
      s[0] += (double)p;
    }
    pnew[idx] = pc;
    vnew[idx] = vn;
  }
  pa_block_reduce_store<1>(s, partials);
}

// s = r - alpha v ; partial sum |s|^2 (tol = |r - alpha v|, linalg.py:230-233)
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_bicg_s(DevGeom G, const SolverScalars* __restrict__ sc,
                                                      const T* __restrict__ r, const T* __restrict__ v,
                                                      T* __restrict__ s_out, double* __restrict__ partials) {
  if (sc->done) return;
  const T alpha = (T)sc->alpha;
  double s[1] = {0.0};
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < G.ncell;
       idx += (int64_t)gridDim.x * blockDim.x) {
    T av = alpha * v[idx];
    T sv = r[idx] - av;
    s_out[idx] = sv;
    T p = sv * sv;
    s[0] += (double)p;
  }
  pa_block_reduce_store<1>(s, partials);
}

// t = A s on S ; partial sums t.s, t.t, r0.t
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_bicg_t(DevGeom G, DevEq<T> E, const SolverScalars* __restrict__ sc,
                                                      Vec<T> sv, const T* __restrict__ r0, T* __restrict__ t_out,
                                                      double* __restrict__ partials) {
  if (sc->done || sc->finished_early) return;
  FieldAcc<T> acc{sv};
  double s[3] = {0.0, 0.0, 0.0};
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < G.ncell;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t i, j, k;
    pa_decode(G, idx, i, j, k);
    T tv = (T)0;
    if (pa_in_S(G, i, j, k)) {
      T sc_ = sv.p[idx];
      tv = pa_apply_terms<T>(G, E, acc, i, j, k, sc_);
      T a = tv * sc_;
      T b = tv * tv;
      T c = r0[idx] * tv;
      s[0] += (double)a;
      s[1] += (double)b;
      s[2] += (double)c;
    }
    t_out[idx] = tv;
  }
  pa_block_reduce_store<3>(s, partials);
}

// early exit: x += alpha p ; otherwise x = x + alpha p + s omega ; r = s - omega t ; |r|^2
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_bicg_x(DevGeom G, const SolverScalars* __restrict__ sc,
                                                      T* __restrict__ x, const T* __restrict__ p,
                                                      const T* __restrict__ s_in, const T* __restrict__ t_in,
                                                      T* __restrict__ r, double* __restrict__ partials,
                                                      const double* pre_part, int pre_n, SolverScalars* sc_w) {
  const T alpha = (T)sc->alpha;
  T omega;
  int early;
  if (pre_n > 0) {
    // folded k_bicg_post stage 12 (rows {|s|^2, t.s, t.t, r0.t} of the fused s / t kernel): stop test 1,
    // then omega and rho_next -- every block on its own, same summation order; block 0 stores
    __shared__ double pre_sm[24];
    const int done_in = sc->done;
    const double tol_lim = sc->tolerance, omega_in = sc->omega;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < pre_n; b += PA_BLOCK) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] += pre_part[4 * (int64_t)b + q];
    }
    if (done_in) return;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      for (int off = 32; off > 0; off >>= 1) v[q] += __shfl_down(v[q], off, 64);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) pre_sm[4 * (threadIdx.x >> 6) + q] = v[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double t4[4] = {0.0, 0.0, 0.0, 0.0};
      for (int w = 0; w < PA_BLOCK / 64; ++w) {
#pragma unroll
        for (int q = 0; q < 4; ++q) t4[q] += pre_sm[4 * w + q];
      }
      const T tol = (T)sqrt(t4[0]);
      const bool bad = isnan(tol) || isinf(tol);
      const int fe = (!bad && (double)tol <= tol_lim) ? 1 : 0;
      T om = (T)omega_in;
      if (!bad && !fe) om = (T)pa_nan_to_num<T>((T)t4[1] / (T)t4[2]);
      pre_sm[16] = (double)om;
      pre_sm[17] = fe ? 1.0 : 0.0;
      pre_sm[18] = bad ? 1.0 : 0.0;
      if (blockIdx.x == 0) {
        sc_w->tol = (double)tol;
        if (bad) {
          sc_w->err = 1;
          sc_w->done = 1;
        } else {
          sc_w->finished_early = fe;
          if (!fe) {
            sc_w->omega = (double)om;
            T rn = -om;
            rn = rn * (T)t4[3];
            sc_w->rho_next = (double)rn;
          }
        }
      }
    }
    __syncthreads();
    if (pre_sm[18] != 0.0) return;
    omega = (T)pre_sm[16];
    early = pre_sm[17] != 0.0;
  } else {
    if (sc->done) return;
    omega = (T)sc->omega;
    early = sc->finished_early;
  }
  double s[1] = {0.0};
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < G.ncell;
       idx += (int64_t)gridDim.x * blockDim.x) {
    T ap = alpha * p[idx];
    T xn = x[idx] + ap;
    if (!early) {
      T so = s_in[idx] * omega;
      xn = xn + so;
      T ot = omega * t_in[idx];
      T rn = s_in[idx] - ot;
      r[idx] = rn;
      T q = rn * rn;
      s[0] += (double)q;
    }
    x[idx] = xn;
  }
  pa_block_reduce_store<1>(s, partials);
}

// stage: 0 after pv (alpha), 1 after s (tol check 1), 2 after t (omega, rho_next), 3 after x (tol check 2)
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_bicg_post(SolverScalars* sc, const double* partials, int nblk,
                                                         int stage) {
  __shared__ double sm[PA_BLOCK / 64];
  if (sc->done) return;
  if (stage == 0) {
    double v = pa_reduce_partials(partials, nblk, 1, 0, sm);
    if (threadIdx.x == 0) {
      sc->itr += 1;
      T r0v = (T)v;
      T rho = (T)sc->rho;
      sc->alpha = pa_nan_to_num<T>(rho / r0v);
    }
  } else if (stage == 1) {
    double v = pa_reduce_partials(partials, nblk, 1, 0, sm);
    if (threadIdx.x == 0) {
      T tol = (T)sqrt(v);
      sc->tol = (double)tol;
      if (isnan(tol) || isinf(tol)) { sc->err = 1; sc->done = 1; return; }
      sc->finished_early = (sc->tol <= sc->tolerance) ? 1 : 0;
    }
  } else if (stage == 2) {
    if (sc->finished_early) return;
    double ts = pa_reduce_partials(partials, nblk, 3, 0, sm);
    double tt = pa_reduce_partials(partials, nblk, 3, 1, sm);
    double r0t = pa_reduce_partials(partials, nblk, 3, 2, sm);
    if (threadIdx.x == 0) {
      T om = (T)pa_nan_to_num<T>((T)ts / (T)tt);
      sc->omega = (double)om;
      T rn = -om;
      rn = rn * (T)r0t;
      sc->rho_next = (double)rn;
    }
  } else if (stage == 12) {
    // fused s / t kernel: partial rows are {|s|^2, t.s, t.t, r0.t}: stop test 1, then omega, rho_next
    double ss = pa_reduce_partials(partials, nblk, 4, 0, sm);
    double ts = pa_reduce_partials(partials, nblk, 4, 1, sm);
    double tt = pa_reduce_partials(partials, nblk, 4, 2, sm);
    double r0t = pa_reduce_partials(partials, nblk, 4, 3, sm);
    if (threadIdx.x == 0) {
      T tol = (T)sqrt(ss);
      sc->tol = (double)tol;
      if (isnan(tol) || isinf(tol)) { sc->err = 1; sc->done = 1; return; }
      sc->finished_early = (sc->tol <= sc->tolerance) ? 1 : 0;
      if (!sc->finished_early) {
        T om = (T)pa_nan_to_num<T>((T)ts / (T)tt);
        sc->omega = (double)om;
        T rn = -om;
        rn = rn * (T)r0t;
        sc->rho_next = (double)rn;
      }
    }
  } else {
    double v = pa_reduce_partials(partials, nblk, 1, 0, sm);
    if (threadIdx.x == 0) {
      if (sc->finished_early) { sc->done = 1; return; }
      T tol = (T)sqrt(v);
      sc->tol = (double)tol;
      if (isnan(tol) || isinf(tol)) { sc->err = 1; sc->done = 1; return; }
      if (sc->tol <= sc->tolerance) sc->done = 1;
      if (sc->itr >= sc->max_it) sc->done = 1;
      // next iteration's beta = rho_next / rho * alpha / omega ; rho = rho_next (linalg.py:212-214)
      T b = (T)sc->rho_next / (T)sc->rho;
      b = b * (T)sc->alpha;
      b = b / (T)sc->omega;
      sc->beta = (double)b;
      sc->rho = sc->rho_next;
    }
  }
}

// ============================================================================
//  host side
// ============================================================================

static thread_local char g_create_err[512] = "";

// stream for the BC / shell helpers: the ctx stream, or the side stream while overlapping
static inline hipStream_t pa_ls(const pa_ctx* c) { return c->launch_stream ? c->launch_stream : c->stream; }

static inline const int* pa_done_flag(const pa_ctx* c) { return &c->sc->done; }

void pa_set_err(pa_ctx* c, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  if (c) vsnprintf(c->err, sizeof(c->err), fmt, ap);
  else vsnprintf(g_create_err, sizeof(g_create_err), fmt, ap);
  va_end(ap);
}

int pa_hip_fail(pa_ctx* c, hipError_t e, const char* what) {
  pa_set_err(c, "HIP error in %s: %s", what, hipGetErrorString(e));
  return PA_E_HIP;
}

int pa_grid_blocks(int64_t work) {
  int64_t b = (work + PA_BLOCK - 1) / PA_BLOCK;
  if (b < 1) b = 1;
  if (b > PA_MAX_GRID) b = PA_MAX_GRID;
  return (int)b;
}

int pa_scratch(pa_ctx* c, void** slot, size_t* cap, size_t bytes) {
  if (*cap >= bytes && *slot) return PA_OK;
  const int q = (int)(slot - c->scr);  // every caller passes &c->scr[id]
  if (*slot) { (void)hipFree(c->scr_base[q]); c->scr_base[q] = nullptr; *slot = nullptr; *cap = 0; }
  if (bytes == 0) return PA_OK;
  // hipMalloc returns 2 MB-aligned blocks; fields that are streamed in lockstep (r, d, d') would then
  // sit at the same offset of their pages at every moment.  Slot q starts q * stagger bytes in.
  static long stagger = -1;
  if (stagger < 0) { const char* e = getenv("PYAPES_HIP_STAGGER"); stagger = e ? atol(e) & ~255L : 0; }
  const size_t off = (size_t)q * (size_t)stagger;
  void* base = nullptr;
  PA_HIP(c, hipMalloc(&base, bytes + off));
  c->scr_base[q] = base;
  *slot = (char*)base + off;
  *cap = bytes;
  return PA_OK;
}

template <typename T>
static void fill_coefs(const pa_ctx* c, DevEq<T>& E) {
  for (int a = 0; a < 3; ++a) {
    T h = (T)c->dx[a];
    T h2 = h * h;
    E.lap.inv[a] = (T)1 / h2;
    E.lap.m2inv[a] = (T)-2 / h2;
    E.lap.c23[a] = (T)(2.0 / 3.0) / h2;
    T th = (T)2 * h;
    T third = (T)(1.0 / 3.0);
    E.grd.h[a] = h;
    E.grd.ih[a] = (T)1 / h;
    E.grd.h2[a] = th;
    E.grd.g[a] = (T)1 / th;
    E.grd.mg[a] = (T)-1 / th;
    T v = (T)1 + third;
    E.grd.lo_p[a] = v / th;
    v = (T)0 - third;
    E.grd.lo_c[a] = v / th;
    v = (T)0 + third;
    E.grd.hi_c[a] = v / th;
    v = (T)-1 - third;
    E.grd.hi_m[a] = v / th;
  }
}

template <typename T>
void pa_build_eq(const pa_ctx* c, int nterms, const pa_term* terms, DevEq<T>& E) {
  memset(&E, 0, sizeof(E));
  E.nterms = nterms;
  for (int q = 0; q < nterms; ++q) {
    E.t[q].kind = terms[q].kind;
    E.t[q].has_coeff = terms[q].has_coeff;
    E.t[q].sign = (T)terms[q].sign;
    E.t[q].coeff = (T)terms[q].coeff;
    E.t[q].coeff_f = (const T*)terms[q].coeff_field;
    E.t[q].u = (T)terms[q].u;
    E.t[q].u_f = (const T*)terms[q].u_field;
  }
  fill_coefs<T>(c, E);
  E.rz = c->coord == PA_COORD_RZ ? (const T*)c->rz_tab : nullptr;
  E.rz_n = c->G.n1;
}
template void pa_build_eq<float>(const pa_ctx*, int, const pa_term*, DevEq<float>&);
template void pa_build_eq<double>(const pa_ctx*, int, const pa_term*, DevEq<double>&);

// interior set, BC flags (mesh/tools.py:7-20, bcs.py:158-163)
void pa_refresh_geom(pa_ctx* c) {
  DevGeom& G = c->G;
  for (int a = 0; a < 3; ++a) {
    int64_t N = a == 0 ? G.g0 : (a == 1 ? G.n1 : G.n2);
    if (!G.act[a]) { G.slo[a] = 0; G.shi[a] = 0; continue; }
    G.slo[a] = 1;
    G.shi[a] = N - 2;
    int lo_pos = -1, hi_pos = -1;
    for (int w = 0; w < c->nbc; ++w) {
      if (c->bc_order[w] == 2 * a) lo_pos = w;
      if (c->bc_order[w] == 2 * a + 1) hi_pos = w;
    }
    G.hi_last[a] = hi_pos > lo_pos;
  }
  for (int f = 0; f < 6; ++f) {
    int t = c->bc[f].type;
    G.bct[f] = t;
    G.treat[f] = (t == PA_BC_NEUMANN || t == PA_BC_SYMMETRY);
    if (t == PA_BC_PERIODIC) {
      int a = f >> 1;
      int64_t N = a == 0 ? G.g0 : (a == 1 ? G.n1 : G.n2);
      if ((f & 1) == 0) G.slo[a] = 0; else G.shi[a] = N - 1;
    }
  }
}

template <typename T>
Vec<T> pa_vec_self(const pa_ctx* c, const T* p) {
  // P = 1 (or no exchange needed): ghost planes are the field's own wrap-around planes
  Vec<T> v;
  v.p = p;
  v.glo = p + (c->G.n0 - 1) * c->G.s0;
  v.ghi = p;
  return v;
}
template Vec<float> pa_vec_self<float>(const pa_ctx*, const float*);
template Vec<double> pa_vec_self<double>(const pa_ctx*, const double*);

template <typename T>
static int bc_apply_t(pa_ctx* c, T* x, bool guarded = false) {
  const DevGeom& G = c->G;
  for (int w = 0; w < c->nbc; ++w) {
    int f = c->bc_order[w];
    const HostBC& b = c->bc[f];
    if (b.type == PA_BC_NONE) continue;
    int a = f >> 1, side = f & 1;
    if (!G.act[a]) continue;
    if (a == 0) {  // slab: only the rank holding the global boundary plane
      if (side == 0 && G.off0 != 0) continue;
      if (side == 1 && G.off0 + G.n0 != G.g0) continue;
    }
    BCArgs<T> B;
    memset(&B, 0, sizeof(B));
    B.axis = a; B.side = side; B.type = b.type;
    B.vals = (const T*)b.vals;
    B.c43 = (T)(4.0 / 3.0);
    B.c13 = (T)(1.0 / 3.0);
    B.c23 = (T)(2.0 / 3.0);
    B.dxf = (T)b.dxf;
    B.ndir = side == 0 ? (T)-1 : (T)1;
    B.done = guarded ? pa_done_flag(c) : nullptr;
    if (b.type == PA_BC_DIRICHLET) B.sval = (T)b.value;
    if (b.type == PA_BC_NEUMANN) {
      // scalar V: python computes 2/3*V in double, casts to the tensor dtype when it meets dx
      T pre = (T)((2.0 / 3.0) * b.value);
      pre = pre * B.dxf;
      pre = pre * B.ndir;
      B.sval = pre;
    }
    if (b.type == PA_BC_PERIODIC && a == 0 && G.n0 != G.g0) {
      B.far0 = (const T*)(side == 0 ? c->bc_far_lo0 : c->bc_far_hi0);
      B.far1 = (const T*)c->bc_far_lo1;
      if (!B.far0) { pa_set_err(c, "periodic axis-0 BC on a slab needs pa_bc_halo_ptrs planes"); return PA_E_STATE; }
    }
    int64_t nu = (a == 0) ? G.n1 : G.n0;
    int64_t nv = (a == 2) ? G.n1 : G.n2;
    hipLaunchKernelGGL(k_bc_face<T>, dim3(pa_grid_blocks(nu * nv)), dim3(PA_BLOCK), 0, pa_ls(c), G, x, B);
  }
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

static bool bc_fusable(const pa_ctx* c);
template <typename T>
static int bc_shell_fused(pa_ctx* c, T* x, double* part2, int with_delta, bool guarded, int* nsh, bool standalone);

static bool bc_pairable(const pa_ctx* c);
template <typename T>
static int bc_pair_apply(pa_ctx* c, T* x, double* part2, int mode, bool guarded, int* nsh,
                         const double* partB = nullptr, int nB = 0, int tail_logic = -1);

// fewest launches that keep the sequential semantics: closed form (2) for small shells, one launch per
// axis (<= 3) for the factory order, else one per face in list order
template <typename T>
static int bc_apply_auto(pa_ctx* c, T* x, bool guarded) {
  if (bc_fusable(c)) return bc_shell_fused<T>(c, x, nullptr, 0, guarded, nullptr, true);
  if (bc_pairable(c)) return bc_pair_apply<T>(c, x, nullptr, 0, guarded, nullptr);
  return bc_apply_t<T>(c, x, guarded);
}

int pa_bc_apply_any(pa_ctx* c, void* x) {
  return c->dtype == PA_F64 ? bc_apply_auto<double>(c, (double*)x, false) : bc_apply_auto<float>(c, (float*)x, false);
}

// ---------------------------------------------------------------------------------------
extern "C" {

const char* pa_version(void) { return "pyapes_hip 0.1 (gfx950)"; }

const char* pa_last_error(const pa_ctx* c) { return c ? c->err : g_create_err; }

int pa_ctx_create(int device, void* hip_stream, pa_ctx** out) {
  if (!out) { pa_set_err(nullptr, "pa_ctx_create: out is NULL"); return PA_E_ARG; }
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    pa_set_err(nullptr, "pa_ctx_create: no HIP device visible (%s)", hipGetErrorString(e));
    return PA_E_HIP;
  }
  if (device < 0 || device >= ndev) { pa_set_err(nullptr, "pa_ctx_create: bad device %d", device); return PA_E_ARG; }
  e = hipSetDevice(device);
  if (e != hipSuccess) { pa_set_err(nullptr, "hipSetDevice: %s", hipGetErrorString(e)); return PA_E_HIP; }
  pa_ctx* c = new (std::nothrow) pa_ctx();
  if (!c) { pa_set_err(nullptr, "out of host memory"); return PA_E_ARG; }
  c->device = device;
  c->stream = (hipStream_t)hip_stream;
  c->err[0] = 0;
  if (const char* fp = getenv("PYAPES_HIP_FASTPATH")) c->fastpath = atoi(fp) != 0;
  if (hipMalloc((void**)&c->sc_base, 2 * sizeof(SolverScalars)) != hipSuccess ||
      hipMalloc((void**)&c->sums, PA_NSUM * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&c->tickets, 4 * sizeof(unsigned int)) != hipSuccess ||
      hipHostMalloc((void**)&c->h_sc, sizeof(SolverScalars)) != hipSuccess ||
      hipHostMalloc((void**)&c->h_poll[0], sizeof(SolverScalars)) != hipSuccess ||
      hipHostMalloc((void**)&c->h_poll[1], sizeof(SolverScalars)) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_poll[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_poll[1], hipEventDisableTiming) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    pa_set_err(nullptr, "pa_ctx_create: allocation failed");
    delete c;
    return PA_E_HIP;
  }
  c->sc = c->sc_base;
  c->sc_alt = c->sc_base + 1;
  (void)hipMemsetAsync(c->sc_base, 0, 2 * sizeof(SolverScalars), c->stream);
  if (const char* fo = getenv("PYAPES_HIP_FOLD")) c->fold = atoi(fo) != 0;
  (void)hipMemsetAsync(c->sums, 0, PA_NSUM * sizeof(double), c->stream);
  (void)hipMemsetAsync(c->tickets, 0, 4 * sizeof(unsigned int), c->stream);
  if (const char* ep = getenv("PYAPES_HIP_EPILOGUE")) c->epilogue = atoi(ep) != 0;
  *out = c;
  return PA_OK;
}

int pa_ctx_destroy(pa_ctx* c) {
  if (!c) return PA_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)pa_comm_destroy(c);
  for (int q = 0; q < PA_NSCRATCH; ++q)
    if (c->scr_base[q]) (void)hipFree(c->scr_base[q]);
  if (c->sc_base) (void)hipFree(c->sc_base);
  if (c->sums) (void)hipFree(c->sums);
  if (c->tickets) (void)hipFree(c->tickets);
  if (c->h_sc) (void)hipHostFree(c->h_sc);
  for (int q = 0; q < 2; ++q) {
    if (c->h_poll[q]) (void)hipHostFree(c->h_poll[q]);
    if (c->ev_poll[q]) (void)hipEventDestroy(c->ev_poll[q]);
  }
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  for (int q = 0; q < 4; ++q)
    if (c->pev[q]) (void)hipEventDestroy(c->pev[q]);
  if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
  if (c->ev_k2) (void)hipEventDestroy(c->ev_k2);
  if (c->ev_bc) (void)hipEventDestroy(c->ev_bc);
  delete c;
  return PA_OK;
}

int pa_grid_set(pa_ctx* c, int ndim, const int64_t* n, const double* dx, int dtype, int64_t i_off,
                int64_t n0_global) {
  if (!c) return PA_E_ARG;
  if (ndim < 1 || ndim > 3) { pa_set_err(c, "pa_grid_set: ndim must be 1..3"); return PA_E_ARG; }
  if (dtype != PA_F32 && dtype != PA_F64) { pa_set_err(c, "pa_grid_set: bad dtype"); return PA_E_ARG; }
  for (int a = 0; a < ndim; ++a)
    if (n[a] < 3 && !(a == 0 && n0_global >= 3 && n[a] >= 1)) {
      pa_set_err(c, "pa_grid_set: every axis needs >= 3 nodes (linalg.py:43-45)");
      return PA_E_ARG;
    }
  if (ndim < 3 && (i_off != 0 || n0_global != n[0])) {
    pa_set_err(c, "pa_grid_set: slab decomposition is for 3-D meshes only (1-D/2-D: replicas)");
    return PA_E_ARG;
  }
  if (i_off < 0 || i_off + n[0] > n0_global) { pa_set_err(c, "pa_grid_set: slab outside the global grid"); return PA_E_ARG; }
  DevGeom& G = c->G;
  memset(&G, 0, sizeof(G));
  int64_t ext[3] = {1, 1, 1};
  double h[3] = {1.0, 1.0, 1.0};
  const int sh = 3 - ndim;
  for (int a = 0; a < ndim; ++a) { ext[a + sh] = n[a]; h[a + sh] = dx[a]; G.act[a + sh] = 1; }
  G.n0 = ext[0]; G.n1 = ext[1]; G.n2 = ext[2];
  G.s1 = G.n2; G.s0 = G.n1 * G.n2;
  G.ncell = G.n0 * G.n1 * G.n2;
  G.off0 = ndim == 3 ? i_off : 0;
  G.g0 = ndim == 3 ? n0_global : 1;
  for (int a = 0; a < 3; ++a) c->dx[a] = h[a];
  c->ndim = ndim;
  c->dtype = dtype;
  c->esize = dtype == PA_F64 ? 8 : 4;
  c->grid_set = 1;
  c->eq_set = 0;
  c->solver_live = 0;
  for (int f = 0; f < 6; ++f) c->bc[f] = HostBC();
  c->nbc = 0;
  c->coord = PA_COORD_XYZ;
  c->rz_tab = nullptr;
  pa_refresh_geom(c);
  return PA_OK;
}

int pa_coord_set(pa_ctx* c, int coord_sys, const void* r_nodes) {
  if (!c || !c->grid_set) { if (c) pa_set_err(c, "pa_coord_set before pa_grid_set"); return PA_E_STATE; }
  if (coord_sys == PA_COORD_XYZ) { c->coord = PA_COORD_XYZ; c->rz_tab = nullptr; return PA_OK; }
  if (coord_sys != PA_COORD_RZ) { pa_set_err(c, "pa_coord_set: unknown coordinate system %d", coord_sys); return PA_E_ARG; }
  if (c->ndim != 2) { pa_set_err(c, "pa_coord_set: rz coordinate system only accepts 2-D grids (_mesh.py:48-49)"); return PA_E_ARG; }
  if (!r_nodes) { pa_set_err(c, "pa_coord_set: rz needs the r coordinates of the nodes"); return PA_E_ARG; }
  PA_HIP(c, hipSetDevice(c->device));
  const int64_t nr = c->G.n1;
  int rc = pa_scratch(c, &c->scr[SCR_RZ], &c->cap[SCR_RZ], (size_t)6 * nr * c->esize);
  if (rc) return rc;
  if (c->dtype == PA_F64)
    hipLaunchKernelGGL(k_rz_tables<double>, dim3(pa_grid_blocks(nr)), dim3(PA_BLOCK), 0, c->stream, nr,
                       (const double*)r_nodes, (double)c->dx[1], (double*)c->scr[SCR_RZ]);
  else
    hipLaunchKernelGGL(k_rz_tables<float>, dim3(pa_grid_blocks(nr)), dim3(PA_BLOCK), 0, c->stream, nr,
                       (const float*)r_nodes, (float)c->dx[1], (float*)c->scr[SCR_RZ]);
  PA_HIP(c, hipGetLastError());
  c->coord = PA_COORD_RZ;
  c->rz_tab = c->scr[SCR_RZ];
  c->solver_live = 0;
  return PA_OK;
}

int pa_bc_clear(pa_ctx* c) {
  if (!c || !c->grid_set) return PA_E_STATE;
  for (int f = 0; f < 6; ++f) c->bc[f] = HostBC();
  c->nbc = 0;
  pa_refresh_geom(c);
  return PA_OK;
}

int pa_bc_set(pa_ctx* c, int face, int order_pos, int type, double value, const void* face_vals, double dxf) {
  if (!c || !c->grid_set) { if (c) pa_set_err(c, "pa_bc_set before pa_grid_set"); return PA_E_STATE; }
  if (face < 0 || face >= 2 * c->ndim) { pa_set_err(c, "pa_bc_set: face %d outside a %d-D mesh", face, c->ndim); return PA_E_ARG; }
  if (order_pos < 0 || order_pos >= 6) { pa_set_err(c, "pa_bc_set: bad order_pos"); return PA_E_ARG; }
  if (type < PA_BC_NONE || type > PA_BC_PERIODIC) { pa_set_err(c, "pa_bc_set: bad type"); return PA_E_ARG; }
  int fi = face + 2 * (3 - c->ndim);
  HostBC& b = c->bc[fi];
  b.type = type; b.value = value; b.vals = face_vals; b.dxf = dxf;
  c->bc_order[order_pos] = fi;
  if (order_pos + 1 > c->nbc) c->nbc = order_pos + 1;
  pa_refresh_geom(c);
  return PA_OK;
}

int pa_apply_bc(pa_ctx* c, void* x) {
  if (!c || !c->grid_set) return PA_E_STATE;
  PA_HIP(c, hipSetDevice(c->device));
  return pa_bc_apply_any(c, x);
}

int pa_eq_set(pa_ctx* c, int nterms, const pa_term* terms) {
  if (!c || !c->grid_set) { if (c) pa_set_err(c, "pa_eq_set before pa_grid_set"); return PA_E_STATE; }
  if (nterms < 1 || nterms > PA_MAX_TERMS) { pa_set_err(c, "pa_eq_set: 1..%d terms", PA_MAX_TERMS); return PA_E_ARG; }
  for (int q = 0; q < nterms; ++q) {
    int k = terms[q].kind;
    if (k < PA_OP_LAPLACIAN || k > PA_OP_DIV_UPWIND) { pa_set_err(c, "pa_eq_set: bad kind"); return PA_E_ARG; }
    if (k == PA_OP_DIV_CENTRAL)
      for (int f = 0; f < 6; ++f)
        if (c->G.treat[f]) {
          pa_set_err(c, "central Div with neumann/symmetry faces: the reference raises IndexError (fdc.py:583)");
          return PA_E_ARG;
        }
    if (k == PA_OP_DIV_CENTRAL && terms[q].u_field && c->G.n0 != c->G.g0 && c->ndim == 3) {
      pa_set_err(c, "tensor-u central Div is single-GPU only");
      return PA_E_ARG;
    }
  }
  c->nterms = nterms;
  for (int q = 0; q < nterms; ++q) c->terms[q] = terms[q];
  c->eq_set = 1;
  return PA_OK;
}

}  // extern "C"

// Grad inside an operator sum only makes sense in 1-D (the reference reshapes the
// (1, mesh.dim, n...) result onto the target, ops.py:145-147)
static int check_eq_applicable(pa_ctx* c) {
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
  if (getenv("PYAPES_HIP_RHS_FULL")) R.nlay = 0;
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

static void pa_profile_stop(pa_ctx* c, int which);

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
    if (!c->x_glo || !c->x_ghi) { pa_set_err(c, "pa_euler_step on a slab needs ghost planes"); return PA_E_STATE; }
    pv.glo = (const T*)c->x_glo; pv.ghi = (const T*)c->x_ghi;
  }
  if (c->profile) (void)hipEventRecord(c->pev[0], c->stream);   // slot 0: the step kernel (without its BC fill)
  int fr = pa_tile3d_euler<T>(c, pv, out, kind, u, u_field, nu, dt);
  if (fr < 0) return fr;
  if (fr == 0)
    hipLaunchKernelGGL(k_euler<T>, dim3(pa_grid_blocks(c->G.ncell)), dim3(PA_BLOCK), 0, c->stream, c->G, El, Ea, pv,
                       out, (T)nu, (T)dt);
  if (c->profile) pa_profile_stop(c, 0);
  PA_HIP(c, hipGetLastError());
  return bc_apply_auto<T>(c, out, false);
}

extern "C" {

int pa_slab_set(pa_ctx* c, const pa_slab* s) {
  if (!c || !c->grid_set) { if (c) pa_set_err(c, "pa_slab_set before pa_grid_set"); return PA_E_STATE; }
  if (c->solver_live) { pa_set_err(c, "pa_slab_set during a solve"); return PA_E_STATE; }
  if (!s) {
    c->slab = 0;
    c->ext_sums = nullptr;
    c->x_glo = c->x_ghi = nullptr;
    c->r_send_lo = c->r_send_hi = nullptr;
    c->r_recv_lo = c->r_recv_hi = nullptr;
    c->bc_far_lo0 = c->bc_far_lo1 = c->bc_far_hi0 = nullptr;
    c->x_pack_lo1 = c->x_pack_hi0 = c->x_pack_hi1 = nullptr;
    return PA_OK;
  }
  if (c->ndim != 3) { pa_set_err(c, "pa_slab_set: slabs are for 3-D meshes"); return PA_E_ARG; }
  if (!s->sums) { pa_set_err(c, "pa_slab_set: sums buffer is required"); return PA_E_ARG; }
  c->slab = 1;
  c->ext_sums = (double*)s->sums;
  c->r_send_lo = s->r_send_lo; c->r_send_hi = s->r_send_hi;
  c->r_recv_lo = s->r_recv_lo; c->r_recv_hi = s->r_recv_hi;
  c->x_glo = s->x_ghost_lo; c->x_ghi = s->x_ghost_hi;
  c->bc_far_lo0 = s->bc_far_lo0; c->bc_far_lo1 = s->bc_far_lo1; c->bc_far_hi0 = s->bc_far_hi0;
  c->x_pack_lo1 = s->x_pack_lo1; c->x_pack_hi0 = s->x_pack_hi0; c->x_pack_hi1 = s->x_pack_hi1;
  return PA_OK;
}

int pa_aop(pa_ctx* c, const void* x, void* y, int interior_only) {
  if (!c || !c->grid_set || !c->eq_set) { if (c) pa_set_err(c, "pa_aop: grid/equation not set"); return PA_E_STATE; }
  if (int rc0 = check_eq_applicable(c)) return rc0;
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
  PA_HIP(c, hipSetDevice(c->device));
  void* buf[2] = {phi, tmp};
  for (int64_t s = 0; s < nsteps; ++s) {
    rc = c->dtype == PA_F64
             ? euler_t<double>(c, (const double*)buf[s & 1], (double*)buf[(s + 1) & 1], kind, u, u_field, nu, dt)
             : euler_t<float>(c, (const float*)buf[s & 1], (float*)buf[(s + 1) & 1], kind, u, u_field, nu, dt);
    if (rc) return rc;
  }
  return PA_OK;
}

}  // extern "C"

// ============================================================================
//  solvers
// ============================================================================
static int shell_blocks(const pa_ctx* c) {
  const DevGeom& G = c->G;
  int64_t tot = 0;
  if (G.act[0]) tot += 2 * G.n1 * G.n2;
  if (G.act[1]) tot += 2 * G.n0 * G.n2;
  if (G.act[2]) tot += 2 * G.n0 * G.n1;
  return pa_grid_blocks(tot);
}
static int64_t shell_elems(const pa_ctx* c) {
  const DevGeom& G = c->G;
  return 2 * (G.n1 * G.n2 + G.n0 * G.n2 + G.n0 * G.n1);
}

// B(x) is a no-op after the first fill when every face is dirichlet (values frozen per solve)
static bool bc_is_static(const pa_ctx* c) {
  for (int f = 0; f < 6; ++f) {
    int t = c->bc[f].type;
    if (t != PA_BC_NONE && t != PA_BC_DIRICHLET) return false;
  }
  return true;
}


// ---- fused BC fill (+ shell stop-test term) ------------------------------------------------------
// usable when the faces are listed in the factory order and every mesh axis has >= 5 nodes
static bool bc_fusable(const pa_ctx* c) {
  if (getenv("PYAPES_HIP_BC_UNFUSED")) return false;
  // Measured on MI355X (512^3 fp64 periodic): the closed form costs 86 + 40 us against 62 + 19 us for
  // six face launches + the shell pass, so it only wins where launches, not bytes, set the time.
  // Against the per-axis pair kernels (explicit Euler step, fp32, us / step fused : pair : faces):
  // 64^3 17 : 20 : 25, 128^3 27.9 : 28.5 : 33, 192^3 44 : 40 : 45, 256^3 65 : 55 : 59 -- the crossover
  // sits between 98 k and 221 k shell nodes.  PYAPES_HIP_BC_FUSED=1 forces the closed form (tests do).
  const int64_t limit = bc_pairable(c) ? 150000 : 400000;
  if (!getenv("PYAPES_HIP_BC_FUSED") &&
      2 * (c->G.n1 * c->G.n2 + c->G.n0 * c->G.n2 + c->G.n0 * c->G.n1) > limit)
    return false;
  int last = -1;
  for (int w = 0; w < c->nbc; ++w) {
    if (c->bc[c->bc_order[w]].type == PA_BC_NONE) continue;
    if (c->bc_order[w] <= last) return false;
    last = c->bc_order[w];
  }
  const DevGeom& G = c->G;
  const int64_t N[3] = {G.g0, G.n1, G.n2};
  const int64_t nloc[3] = {G.n0, G.n1, G.n2};
  for (int a = 0; a < 3; ++a)
    if (G.act[a] && (N[a] < 5 || (a == 0 && nloc[0] < 3))) return false;
  // periodic must be declared on both faces of an axis for the closed form
  for (int a = 0; a < 3; ++a)
    if ((c->bc[2 * a].type == PA_BC_PERIODIC) != (c->bc[2 * a + 1].type == PA_BC_PERIODIC)) return false;
  return true;
}

template <typename T>
static int bc_fill_all(pa_ctx* c, BCAll<T>& B) {
  memset(&B, 0, sizeof(B));
  const DevGeom& G = c->G;
  for (int f = 0; f < 6; ++f) {
    const HostBC& b = c->bc[f];
    BCFaceDev<T>& F = B.f[f];
    F.type = b.type;
    F.vals = (const T*)b.vals;
    F.dxf = (T)b.dxf;
    F.ndir = (f & 1) == 0 ? (T)-1 : (T)1;
    if (b.type == PA_BC_DIRICHLET) F.sval = (T)b.value;
    if (b.type == PA_BC_NEUMANN) {
      T pre = (T)((2.0 / 3.0) * b.value);
      pre = pre * F.dxf;
      pre = pre * F.ndir;
      F.sval = pre;
    }
    if ((f >> 1) == 0 && G.act[0]) {  // slab: a rank only applies the axis-0 face it holds
      if (f == 0 && G.off0 != 0) F.type = PA_BC_NONE;
      if (f == 1 && G.off0 + G.n0 != G.g0) F.type = PA_BC_NONE;
    }
  }
  B.c43 = (T)(4.0 / 3.0);
  B.c13 = (T)(1.0 / 3.0);
  B.c23 = (T)(2.0 / 3.0);
  B.slab_periodic0 = (G.n0 != G.g0 && c->bc[0].type == PA_BC_PERIODIC) ? 1 : 0;
  if (B.slab_periodic0) {
    B.far_lo0 = (const T*)c->bc_far_lo0;
    B.far_lo1 = (const T*)c->bc_far_lo1;
    B.far_hi0 = (const T*)c->bc_far_hi0;
    if ((B.f[0].type == PA_BC_PERIODIC && (!B.far_lo0 || !B.far_lo1)) ||
        (B.f[1].type == PA_BC_PERIODIC && !B.far_hi0)) {
      pa_set_err(c, "periodic axis-0 BC on a slab needs the far planes (pa_slab_set)");
      return PA_E_STATE;
    }
  }
  return PA_OK;
}

// fills x; with_delta: partial sums of (new - old)^2 over the shell -> part2 (returns #blocks via *nsh)
template <typename T>
static int bc_shell_fused(pa_ctx* c, T* x, double* part2, int with_delta, bool guarded, int* nsh,
                          bool standalone) {
  BCAll<T> B;
  int rc = bc_fill_all<T>(c, B);
  if (rc) return rc;
  const int nb = shell_blocks(c);
  const size_t half = (size_t)shell_elems(c);
  T *so, *sn;
  if (standalone) {  // plain pa_apply_bc: private staging, the solver's x_old shell is left alone
    if ((rc = pa_scratch(c, &c->scr[SCR_SHELL2], &c->cap[SCR_SHELL2], half * sizeof(T)))) return rc;
    so = sn = (T*)c->scr[SCR_SHELL2];
  } else {
    T* base = (T*)c->scr[SCR_SHELL];
    so = base + (c->shell_cur ? half : 0);
    sn = base + (c->shell_cur ? 0 : half);
  }
  const int* done = guarded ? pa_done_flag(c) : nullptr;
  // Without a periodic face the closed form only ever READS nodes that no face writes: a face value is a
  // formula over the nodes 1 and 2 (N-2, N-3) steps inside along its axis at the stage before it, and
  // following that down ends at nodes that lie on no face with a BC (every axis has >= 5 nodes).  So the
  // compute kernel may store into x itself and the scatter launch is dropped.  A periodic face reads
  // x[N-1] / x[N-2] raw -- shell nodes other threads write -- and keeps the two passes.
  bool direct = !c->slab && !getenv("PYAPES_HIP_BC_TWO_PASS");
  for (int f = 0; f < 6; ++f)
    if (c->bc[f].type == PA_BC_PERIODIC) direct = false;
  hipLaunchKernelGGL(k_bc_compute<T>, dim3(nb), dim3(PA_BLOCK), 0, pa_ls(c), c->G, B, done, (const T*)x,
                     (const T*)so, sn, part2, with_delta, direct ? x : (T*)nullptr);
  if (!direct)
    hipLaunchKernelGGL(k_bc_scatter<T>, dim3(nb), dim3(PA_BLOCK), 0, pa_ls(c), c->G, done, x, (const T*)sn);
  if (!standalone) c->shell_cur ^= 1;
  if (nsh) *nsh = nb;
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

// epilogue descriptors (pa_epilogue.h); kind 0 = "keep the separate k_cg_post_* launch"
static CgEpi epi_none() {
  CgEpi e;
  memset(&e, 0, sizeof(e));
  return e;
}
static CgEpi epi_a(pa_ctx* c, const double* part, int logic) {
  CgEpi e = epi_none();
  if (!c->epilogue || c->profile) return e;
  e.kind = 1; e.logic = logic; e.ticket = c->tickets + 0; e.sc = c->sc; e.sums = pa_sums(c);
  e.part = part; e.npart = -1;
  return e;
}
static CgEpi epi_b(pa_ctx* c, int slot, const double* part, int npart, const double* part_shell, int nshell, int logic) {
  CgEpi e = epi_none();
  if (!c->epilogue || c->profile) return e;
  e.kind = 2; e.logic = logic; e.ticket = c->tickets + slot; e.sc = c->sc; e.sums = pa_sums(c);
  e.part = part; e.npart = npart; e.part_shell = part_shell; e.nshell = nshell;
  return e;
}

// BC list in factory order with both faces of every mesh axis present (what the BC factories emit),
// >= 5 nodes per axis: the per-axis pair kernels apply
static bool bc_pairable(const pa_ctx* c) {
  if (getenv("PYAPES_HIP_BC_UNPAIRED")) return false;
  int last = -1, cnt = 0;
  for (int w = 0; w < c->nbc; ++w) {
    if (c->bc[c->bc_order[w]].type == PA_BC_NONE) continue;
    if (c->bc_order[w] <= last) return false;
    last = c->bc_order[w];
    ++cnt;
  }
  const DevGeom& G = c->G;
  const int64_t N[3] = {G.g0, G.n1, G.n2};
  int need = 0;
  for (int a = 0; a < 3; ++a) {
    if (!G.act[a]) continue;
    need += 2;
    if (N[a] < 5 || (a == 0 && G.n0 < 3)) return false;
    if (c->bc[2 * a].type == PA_BC_NONE || c->bc[2 * a + 1].type == PA_BC_NONE) return false;
    if ((c->bc[2 * a].type == PA_BC_PERIODIC) != (c->bc[2 * a + 1].type == PA_BC_PERIODIC)) return false;
  }
  return cnt == need;
}

template <typename T>
static void bc_face_args(pa_ctx* c, int f, BCArgs<T>& B, bool guarded) {
  const DevGeom& G = c->G;
  const HostBC& b = c->bc[f];
  memset(&B, 0, sizeof(B));
  const int a = f >> 1, side = f & 1;
  B.axis = a; B.side = side; B.type = b.type;
  if (a == 0) {  // slab: only the rank holding the global boundary plane
    if (side == 0 && G.off0 != 0) B.type = 0;
    if (side == 1 && G.off0 + G.n0 != G.g0) B.type = 0;
  }
  B.vals = (const T*)b.vals;
  B.c43 = (T)(4.0 / 3.0);
  B.c13 = (T)(1.0 / 3.0);
  B.c23 = (T)(2.0 / 3.0);
  B.dxf = (T)b.dxf;
  B.ndir = side == 0 ? (T)-1 : (T)1;
  B.done = guarded ? pa_done_flag(c) : nullptr;
  if (b.type == PA_BC_DIRICHLET) B.sval = (T)b.value;
  if (b.type == PA_BC_NEUMANN) {
    T pre = (T)((2.0 / 3.0) * b.value);
    pre = pre * B.dxf;
    pre = pre * B.ndir;
    B.sval = pre;
  }
  if (b.type == PA_BC_PERIODIC && a == 0 && G.n0 != G.g0) {
    B.far0 = (const T*)(side == 0 ? c->bc_far_lo0 : c->bc_far_hi0);
    B.far1 = (const T*)c->bc_far_lo1;
  }
}

// mode 0: fill only; 1: fill + shell delta (partials -> part2, rows returned in *nsh) + save; 2: fill + save;
// 3: save only (slab: the driver has filled the BCs itself)
template <typename T>
static int bc_pair_apply(pa_ctx* c, T* x, double* part2, int mode, bool guarded, int* nsh,
                         const double* partB, int nB, int tail_logic) {
  const DevGeom& G = c->G;
  const int64_t sz[3] = {G.n1 * G.n2, G.n0 * G.n2, G.n0 * G.n1};
  int64_t start[6], total = 0;
  for (int f = 0; f < 6; ++f) { start[f] = total; total += G.act[f >> 1] ? sz[f >> 1] : 0; }
  T* shell = (T*)c->scr[SCR_SHELL];
  int rows = 0;
  int last_axis = -1;  // the launch whose last block finishes the B-chain reduction (tail_logic >= 0)
  if (tail_logic >= 0 && mode == 1)
    for (int a = 0; a < 3; ++a) {
      if (!G.act[a]) continue;
      BCArgs<T> lo, hi;
      bc_face_args<T>(c, 2 * a, lo, guarded);
      bc_face_args<T>(c, 2 * a + 1, hi, guarded);
      if (lo.type || hi.type) last_axis = a;
    }
  c->b_tail_done = 0;
  for (int a = 0; a < 3; ++a) {
    if (!G.act[a]) continue;
    BCPairArgs<T> P;
    bc_face_args<T>(c, 2 * a, P.lo, guarded);
    bc_face_args<T>(c, 2 * a + 1, P.hi, guarded);
    if (mode != 3 && ((P.lo.type == PA_BC_PERIODIC && a == 0 && G.n0 != G.g0 && !P.lo.far0) ||
                      (P.hi.type == PA_BC_PERIODIC && a == 0 && G.n0 != G.g0 && !P.hi.far0))) {
      pa_set_err(c, "periodic axis-0 BC on a slab needs the far planes (pa_slab_set)");
      return PA_E_STATE;
    }
    if (!P.lo.type && !P.hi.type) continue;
    P.axis = a;
    P.pos_lo = start[2 * a];
    P.pos_hi = start[2 * a + 1];
    const int nb = pa_grid_blocks(sz[a]);
    CgEpi epi = epi_none();
    if (a == last_axis) {
      epi = epi_b(c, 2, partB, nB, part2, rows + nb, tail_logic);
      if (epi.kind) c->b_tail_done = 1;
    }
    hipLaunchKernelGGL(k_bc_pair<T>, dim3(nb), dim3(PA_BLOCK), 0, pa_ls(c), G, x, P,
                       guarded ? pa_done_flag(c) : (const int*)nullptr, shell, part2 ? part2 + rows : nullptr,
                       mode, epi);
    if (mode == 1) rows += nb;
  }
  if (nsh) *nsh = rows;
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

static int init_scalars(pa_ctx* c, double tol, int64_t max_it) {
  SolverScalars h;
  memset(&h, 0, sizeof(h));
  h.tolerance = tol;
  h.max_it = max_it;
  h.tol = 1.0;
  h.rho = 1.0; h.alpha = 1.0; h.omega = 1.0;
  h.done = !(1.0 > tol);  // `while tol > tolerance` with tol = 1.0 (linalg.py:90,109)
  *c->h_sc = h;
  PA_HIP(c, hipMemcpyAsync(c->sc, c->h_sc, sizeof(h), hipMemcpyHostToDevice, c->stream));
  PA_HIP(c, hipStreamSynchronize(c->stream));
  return PA_OK;
}

static int pa_join_side(pa_ctx* c) {
  if (c->side_pending) {
    PA_HIP(c, hipStreamWaitEvent(c->stream, c->ev_bc, 0));
    c->side_pending = 0;
  }
  return PA_OK;
}

static int read_scalars(pa_ctx* c) {
  if (int rcj = pa_join_side(c)) return rcj;
  PA_HIP(c, hipMemcpyAsync(c->h_sc, c->sc, sizeof(SolverScalars), hipMemcpyDeviceToHost, c->stream));
  PA_HIP(c, hipStreamSynchronize(c->stream));
  return PA_OK;
}

// Pipelined poll: after a batch of iterations has been enqueued (and its pending scalar step flushed),
// queue a copy of the device scalars and wait for the copy of the PREVIOUS batch.  The GPU already has
// the next batch to work on while the host looks at the flag; when the flag was set, that batch is
// no-ops (every kernel starts with `if (done) return`), so results and iteration counts are unchanged.
// A synchronous poll leaves the GPU idle for a host round trip (~250 us) every ~300 us of work on the
// meshes of the reference's tests.
struct PollPipe {
  int pending = -1, slot = 0;
};
static int poll_submit(pa_ctx* c, PollPipe& P, bool* done) {
  *done = false;
  if (int rcj = pa_join_side(c)) return rcj;
  PA_HIP(c, hipMemcpyAsync(c->h_poll[P.slot], c->sc, sizeof(SolverScalars), hipMemcpyDeviceToHost, c->stream));
  PA_HIP(c, hipEventRecord(c->ev_poll[P.slot], c->stream));
  if (P.pending >= 0) {
    PA_HIP(c, hipEventSynchronize(c->ev_poll[P.pending]));
    *done = c->h_poll[P.pending]->done != 0;
  }
  P.pending = P.slot;
  P.slot ^= 1;
  return PA_OK;
}
static int poll_drain(pa_ctx* c, PollPipe& P, bool* done) {
  *done = false;
  if (P.pending < 0) return PA_OK;
  PA_HIP(c, hipEventSynchronize(c->ev_poll[P.pending]));
  *done = c->h_poll[P.pending]->done != 0;
  P.pending = -1;
  return PA_OK;
}

static int poll_interval(const pa_ctx* c) {
  // keep >= ~300 us of queued GPU work between host polls of the done flag
  double est_us = (double)c->G.ncell * 80.0 / 4.0e6 + 30.0;
  int k = (int)ceil(300.0 / est_us);
  return std::max(1, std::min(k, 64));
}

static void fill_report(pa_ctx* c, pa_report* out, float ms) {
  const SolverScalars& h = *c->h_sc;
  out->itr = h.itr;
  out->tol = h.tol;
  out->converge = h.itr < h.max_it;
  out->status = h.err ? PA_E_NONFINITE : PA_OK;
  out->rr = h.rr;
  out->gpu_ms = ms;
}

// r = (b - A x) on S (0 elsewhere), d = r, per-block partial sums of r.r: the tiled A x kernel plus one
// streaming pass where the tiled kernel applies, else the generic kernel
template <typename T>
static int cg_residual_init(pa_ctx* c, const DevEq<T>& E, Vec<T> xv, const T* rhs, T* r, T* d, T* send_lo,
                            T* send_hi, double* part) {
  const int nblk = pa_grid_blocks(c->G.ncell);
  // slab: a NULL ghost plane marks a physical (non-periodic) end.  No result ever uses that plane (the
  // end plane is a boundary node, outside S), but the tiled kernel loads it speculatively: the field's
  // own end plane stands in, so the load stays inside valid memory.
  Vec<T> xt = xv;
  if (!xt.glo) xt.glo = xt.p;
  if (!xt.ghi) xt.ghi = xt.p + (c->G.n0 - 1) * c->G.s0;
  int fr = (rhs != r && (const T*)xv.p != r) ? pa_tile3d_aop<T>(c, E, xt, r, 1) : 0;
  if (fr < 0) return fr;
  if (fr > 0)
    hipLaunchKernelGGL(k_cg_init_ax<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, c->G, rhs, r, d, send_lo, send_hi,
                       part);
  else
    hipLaunchKernelGGL(k_cg_init<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, c->G, E, xv, rhs, r, d, send_lo,
                       send_hi, part);
  return PA_OK;
}

template <typename T>
static int cg_begin_t(pa_ctx* c, T* x, const T* rhs, double tol, int64_t max_it) {
  const DevGeom& G = c->G;
  const size_t fb = (size_t)G.ncell * sizeof(T);
  const int nblk = pa_grid_blocks(G.ncell);
  int rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_R], &c->cap[SCR_R], fb))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_D0], &c->cap[SCR_D0], fb))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_D1], &c->cap[SCR_D1], fb))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_PART], &c->cap[SCR_PART], (size_t)PA_MAX_PARTIALS * 4 * sizeof(double)))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_PART2], &c->cap[SCR_PART2], (size_t)3 * PA_MAX_GRID * sizeof(double)))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_SHELL], &c->cap[SCR_SHELL], 2 * (size_t)shell_elems(c) * sizeof(T)))) return rc;
  if ((rc = init_scalars(c, tol, max_it))) return rc;
  // the tiled phase kernels do not visit the last boundary row / column of non-periodic axes: the
  // direction there is 0 by definition and has to be 0 in the buffer the first phase A writes into
  PA_HIP(c, hipMemsetAsync(c->scr[SCR_D1], 0, fb, c->stream));
  c->cg_x = x;
  c->cur = 0;
  c->fold_a_n = c->fold_b_n = c->fold_b_nsh = 0;  // nothing of an earlier (possibly failed) solve is pending
  c->bc_static = bc_is_static(c);
  c->bc_fused = bc_fusable(c);
  c->bc_pair = (!c->bc_fused && bc_pairable(c)) ? 1 : 0;
  c->shell_cur = 0;
  c->side_pending = 0;
  c->launch_stream = nullptr;
  {
    // Opt-in (PYAPES_HIP_OVERLAP=1).  Measured on MI355X at 512^3 fp64 periodic: no gain (1.72 vs
    // 1.71 ms / iteration) -- the chip is bandwidth-bound either way and the strided face kernels
    // take their HBM time from phase A instead of from the critical path.
    const char* ov = getenv("PYAPES_HIP_OVERLAP");
    c->overlap = (!c->slab && !c->bc_static && ov && atoi(ov) != 0) ? 1 : 0;
    if (c->overlap && !c->side) {
      if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&c->ev_k2, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&c->ev_bc, hipEventDisableTiming) != hipSuccess) {
        c->overlap = 0;
        (void)hipGetLastError();
      }
    }
    if (c->overlap) c->bc_pair = 0;
  }
  c->solver_live = 1;
  DevEq<T> E;
  pa_build_eq<T>(c, c->nterms, c->terms, E);
  // linalg.py:97.  On a slab the driver fills the BCs itself (pa_apply_bc) BEFORE it exchanges
  // the ghost planes of x, so the fill must not run again here.
  bool shell_ready = false;
  if (!c->slab) {
    if (c->bc_fused) {  // fill + remember the filled shell as x_old in one go
      if ((rc = bc_shell_fused<T>(c, x, nullptr, 0, false, nullptr, false))) return rc;
      shell_ready = true;
    } else if (c->bc_pair) {
      if ((rc = bc_pair_apply<T>(c, x, nullptr, 2, false, nullptr))) return rc;
      shell_ready = true;
    } else if ((rc = bc_apply_t<T>(c, x))) {
      return rc;
    }
  } else if (c->bc_pair) {  // slab: the driver has filled the BCs already; only record the shell
    if ((rc = bc_pair_apply<T>(c, x, nullptr, 3, false, nullptr))) return rc;
    shell_ready = true;
  }
  T* r = (T*)c->scr[SCR_R];
  T* d = (T*)c->scr[SCR_D0];
  double* part = (double*)c->scr[SCR_PART];
  Vec<T> xv = pa_vec_self<T>(c, x);
  if (c->slab) { xv.glo = (const T*)c->x_glo; xv.ghi = (const T*)c->x_ghi; }
  if ((rc = cg_residual_init<T>(c, E, xv, rhs, r, d, (T*)c->r_send_lo, (T*)c->r_send_hi, part))) return rc;
  hipLaunchKernelGGL(k_cg_post_init<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, part, nblk, pa_sums(c),
                     c->slab ? 0 : 2);
  c->pending_init_logic = c->slab ? 1 : 0;
  if (c->slab) {
    // ghost planes of the two direction buffers: lo/hi x ping/pong, zero = "d = r" with beta = 0
    const size_t pb = (size_t)G.s0 * sizeof(T);
    if ((rc = pa_scratch(c, &c->scr[SCR_GHOST], &c->cap[SCR_GHOST], 4 * pb))) return rc;
    PA_HIP(c, hipMemsetAsync(c->scr[SCR_GHOST], 0, 4 * pb, c->stream));
    char* g = (char*)c->scr[SCR_GHOST];
    c->d_glo[0] = g; c->d_ghi[0] = g + pb; c->d_glo[1] = g + 2 * pb; c->d_ghi[1] = g + 3 * pb;
  }
  if (!shell_ready)
    hipLaunchKernelGGL(k_shell<T>, dim3(shell_blocks(c)), dim3(PA_BLOCK), 0, c->stream, G, c->sc, (const T*)x,
                       (T*)c->scr[SCR_SHELL] + (c->shell_cur ? shell_elems(c) : 0), (double*)c->scr[SCR_PART2], 0);
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

// profiling build of the launch path: HIP events on the ctx stream bracket exactly one
// dominant kernel; the host waits for each, so use it in a dedicated measurement loop only
static void pa_profile_stop(pa_ctx* c, int which) {
  hipEvent_t e0 = c->pev[2 * which], e1 = c->pev[2 * which + 1];
  (void)hipEventRecord(e1, c->stream);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) {
    c->prof_ms[which] += ms;
    c->prof_n[which] += 1;
  }
}

template <typename T>
static Vec<T> cg_vec(pa_ctx* c, const T* p, int which /*0 r, 1 d cur*/) {
  Vec<T> v = pa_vec_self<T>(c, p);
  if (c->slab) {
    // a NULL recv pointer marks a physical (non-periodic) end: that ghost plane is never used in a
    // result, the field's own plane stands in so that speculative loads stay inside valid memory
    if (which == 0) {
      if (c->r_recv_lo) v.glo = (const T*)c->r_recv_lo;
      if (c->r_recv_hi) v.ghi = (const T*)c->r_recv_hi;
    } else {
      if (c->r_recv_lo) v.glo = (const T*)c->d_glo[c->cur];
      if (c->r_recv_hi) v.ghi = (const T*)c->d_ghi[c->cur];
    }
  }
  return v;
}

// scalar steps that were left to the prologue of a tiled kernel that is not coming (the generic kernel
// runs instead, or the batch of iterations ends): run them as the single-block kernels they replace
template <typename T>
static void cg_flush_fold(pa_ctx* c) {
  if (c->fold_a_n > 0) {
    hipLaunchKernelGGL(k_cg_post_a<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc,
                       (const double*)c->scr[SCR_PART] + 2 * (size_t)PA_MAX_PARTIALS, c->fold_a_n, pa_sums(c), 2);
    c->fold_a_n = 0;
  }
  if (c->fold_b_n > 0) {
    hipLaunchKernelGGL(k_cg_post_b<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, c->fold_b_part,
                       c->fold_b_n, (const double*)c->scr[SCR_PART2], c->fold_b_nsh, pa_sums(c), 2);
    c->fold_b_n = c->fold_b_nsh = 0;
  }
}

template <typename T>
int pa_cg_phase_a_t(pa_ctx* c, int stage_post) {
  const DevGeom& G = c->G;
  const int nblk = pa_grid_blocks(G.ncell);
  DevEq<T> E;
  pa_build_eq<T>(c, c->nterms, c->terms, E);
  T* r = (T*)c->scr[SCR_R];
  T* dold = (T*)c->scr[c->cur ? SCR_D1 : SCR_D0];
  T* dnew = (T*)c->scr[c->cur ? SCR_D0 : SCR_D1];
  double* part = (double*)c->scr[SCR_PART];
  // inside pa_cg_iterate on one GPU the two single-block scalar kernels of an iteration are folded into
  // the prologue of the tiled kernel that follows them (pa_cg3d_kernel.h); d.Ad rows then live in the
  // upper half of SCR_PART, because phase B writes its own rows while its blocks still read these
  const bool foldable = c->fold && c->in_iterate && stage_post == 2 && !c->slab && !c->overlap && !c->profile &&
                        !c->epilogue;
  if (foldable) part += 2 * (size_t)PA_MAX_PARTIALS;
  if (c->pending_init_logic) {  // slab: sum r.r has been all-reduced by the driver
    hipLaunchKernelGGL(k_cg_post_init<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, (const double*)nullptr, 0,
                       pa_sums(c), 1);
    c->pending_init_logic = 0;
  }
  Vec<T> rv = cg_vec<T>(c, r, 0), dv = cg_vec<T>(c, dold, 1);
  if (c->slab && (c->r_recv_lo || c->r_recv_hi)) {
    hipLaunchKernelGGL(k_ghost_dir<T>, dim3(pa_grid_blocks(G.s0)), dim3(PA_BLOCK), 0, c->stream, c->sc, G.s0,
                       (const T*)c->r_recv_lo, (const T*)c->r_recv_hi, (const T*)c->d_glo[c->cur],
                       (const T*)c->d_ghi[c->cur], (T*)c->d_glo[c->cur ^ 1], (T*)c->d_ghi[c->cur ^ 1]);
  }
  if (c->profile) (void)hipEventRecord(c->pev[0], c->stream);
  const CgEpi epa = epi_a(c, part, stage_post == 2);
  int rc = pa_cg3d_phase_a<T>(c, E, rv, dv, dnew, part, epa);
  if (rc < 0) return rc;
  int used_blocks = rc;
  if (rc == 0) {
    cg_flush_fold<T>(c);  // the tiled kernel declined: the previous iteration is closed by its own kernel
    hipLaunchKernelGGL(k_cg_a<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, G, E, c->sc, rv, dv, dnew, part, epa);
    used_blocks = nblk;
  }
  if (c->profile) pa_profile_stop(c, 0);
  c->cur ^= 1;
  if (foldable && !epa.kind && used_blocks <= PA_MAX_GRID)
    c->fold_a_n = used_blocks;  // phase B's prologue (or cg_flush_fold) computes alpha
  else if (!epa.kind)
    hipLaunchKernelGGL(k_cg_post_a<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, part, used_blocks, pa_sums(c),
                       stage_post);
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

template <typename T>
int pa_cg_phase_b_t(pa_ctx* c, int stage_post) {
  const DevGeom& G = c->G;
  const int nblk = pa_grid_blocks(G.ncell);
  DevEq<T> E;
  pa_build_eq<T>(c, c->nterms, c->terms, E);
  T* r = (T*)c->scr[SCR_R];
  T* d = (T*)c->scr[c->cur ? SCR_D1 : SCR_D0];
  T* x = (T*)c->cg_x;
  double* part = (double*)c->scr[SCR_PART];
  double* part2 = (double*)c->scr[SCR_PART2];
  Vec<T> dv = cg_vec<T>(c, d, 1);
  if (int rcj = pa_join_side(c)) return rcj;  // x and the done flag of the previous iteration are final
  if (c->profile) (void)hipEventRecord(c->pev[2], c->stream);
  // with frozen (all-dirichlet) BCs nothing follows phase B: its last block finishes the iteration
  CgEpi epb = epi_none();
  if (c->bc_static && !(c->overlap && stage_post == 2)) epb = epi_b(c, 1, part, -1, nullptr, 0, stage_post == 2);
  c->b_tail_done = epb.kind ? 1 : 0;
  int rc = pa_cg3d_phase_b<T>(c, E, dv, x, r, part, epb);
  if (rc < 0) return rc;
  int used_blocks = rc;
  if (rc == 0) {
    cg_flush_fold<T>(c);  // alpha by its own kernel
    hipLaunchKernelGGL(k_cg_b<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, G, E, c->sc, dv, x, r,
                       (T*)c->r_send_lo, (T*)c->r_send_hi, part, epb);
    used_blocks = nblk;
  }
  if (c->profile) pa_profile_stop(c, 1);
  c->b_blocks = used_blocks;
  if (c->slab) {  // BC fill + shell + reduction happen in pa_cg_bc, after the driver's plane exchange
    if (c->x_pack_lo1 || c->x_pack_hi0 || c->x_pack_hi1) {
      const T* xr = (const T*)x;
      hipLaunchKernelGGL(k_pack_planes<T>, dim3(pa_grid_blocks(G.s0)), dim3(PA_BLOCK), 0, c->stream, c->sc, G.s0,
                         xr + 1 * G.s0, (T*)c->x_pack_lo1, xr + (G.n0 - 1) * G.s0, (T*)c->x_pack_hi0,
                         xr + (G.n0 - 2) * G.s0, (T*)c->x_pack_hi1);
    }
    PA_HIP(c, hipGetLastError());
    return PA_OK;
  }
  if (c->overlap && stage_post == 2) {
    // main stream: beta for the next phase A.  side stream: BC fill, shell term, stop test -- they
    // overlap phase A of the next iteration and are joined in front of its phase B / any read-back.
    hipLaunchKernelGGL(k_cg_post_b1<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, part, used_blocks, pa_sums(c));
    PA_HIP(c, hipEventRecord(c->ev_k2, c->stream));
    PA_HIP(c, hipStreamWaitEvent(c->side, c->ev_k2, 0));
    c->launch_stream = c->side;
    int nsh2 = 0;
    if (c->bc_fused) {
      rc = bc_shell_fused<T>(c, x, part2, 1, true, &nsh2, false);
    } else {
      rc = bc_apply_t<T>(c, x, true);
      nsh2 = shell_blocks(c);
      if (!rc)
        hipLaunchKernelGGL(k_shell<T>, dim3(nsh2), dim3(PA_BLOCK), 0, c->side, G, c->sc, (const T*)x,
                           (T*)c->scr[SCR_SHELL], part2, 1);
    }
    c->launch_stream = nullptr;
    if (rc) return rc;
    hipLaunchKernelGGL(k_cg_post_b2<T>, dim3(1), dim3(PA_BLOCK), 0, c->side, c->sc, (const double*)part2, nsh2,
                       (const double*)pa_sums(c));
    PA_HIP(c, hipEventRecord(c->ev_bc, c->side));
    c->side_pending = 1;
    PA_HIP(c, hipGetLastError());
    return PA_OK;
  }
  int nsh = 0;
  if (!c->bc_static) {
    if (c->bc_fused) {
      if ((rc = bc_shell_fused<T>(c, x, part2, 1, true, &nsh, false))) return rc;
    } else if (c->bc_pair) {
      if ((rc = bc_pair_apply<T>(c, x, part2, 1, true, &nsh, part, used_blocks, stage_post == 2))) return rc;
    } else {
      if ((rc = bc_apply_t<T>(c, x, true))) return rc;
      nsh = shell_blocks(c);
      hipLaunchKernelGGL(k_shell<T>, dim3(nsh), dim3(PA_BLOCK), 0, c->stream, G, c->sc, (const T*)x,
                         (T*)c->scr[SCR_SHELL], part2, 1);
    }
  }
  const bool foldable = c->fold && c->in_iterate && stage_post == 2 && !c->slab && !c->profile && !c->epilogue &&
                        used_blocks <= PA_MAX_GRID && nsh <= 3 * PA_MAX_GRID;
  if (!c->b_tail_done && foldable) {
    c->fold_b_n = used_blocks;  // the next phase A's prologue (or cg_flush_fold) closes this iteration
    c->fold_b_nsh = nsh;
    c->fold_b_part = part;
  } else if (!c->b_tail_done) {
    hipLaunchKernelGGL(k_cg_post_b<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, part, used_blocks, part2, nsh,
                       pa_sums(c), stage_post);
  }
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

// slab: BC fill of x (needs the far planes the driver just exchanged when axis 0 is periodic),
// boundary-shell part of the stop test, local partial sums -> sums[1], sums[2]
template <typename T>
int pa_cg_bc_t(pa_ctx* c) {
  const DevGeom& G = c->G;
  T* x = (T*)c->cg_x;
  double* part = (double*)c->scr[SCR_PART];
  double* part2 = (double*)c->scr[SCR_PART2];
  int nsh = 0, rc;
  if (!c->bc_static) {
    if (c->bc_fused) {
      if ((rc = bc_shell_fused<T>(c, x, part2, 1, true, &nsh, false))) return rc;
    } else if (c->bc_pair) {
      if ((rc = bc_pair_apply<T>(c, x, part2, 1, true, &nsh, part, c->b_blocks, 0))) return rc;
    } else {
      if ((rc = bc_apply_t<T>(c, x, true))) return rc;
      nsh = shell_blocks(c);
      hipLaunchKernelGGL(k_shell<T>, dim3(nsh), dim3(PA_BLOCK), 0, c->stream, G, c->sc, (const T*)x,
                         (T*)c->scr[SCR_SHELL], part2, 1);
    }
  }
  if (!c->b_tail_done)   // (frozen BCs: phase B's own epilogue has reduced already)
    hipLaunchKernelGGL(k_cg_post_b<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, part, c->b_blocks, part2, nsh,
                       pa_sums(c), 0);
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

template <typename T>
static int cg_run_t(pa_ctx* c, T* x, const T* rhs, double tol, int64_t max_it, pa_report* out) {
  int rc = cg_begin_t<T>(c, x, rhs, tol, max_it);
  if (rc) return rc;
  const int poll = poll_interval(c);
  PA_HIP(c, hipEventRecord(c->ev0, c->stream));
  int64_t enq = 0;
  c->in_iterate = 1;  // scalar steps folded into the next tiled kernel's prologue (flushed before every poll)
  PollPipe P;
  bool done = false;
  int64_t batch = 1;
  while (!done && !rc) {
    // the device stops by itself after max_it + 1 iterations (linalg.py K+1 quirk): never enqueue more
    int64_t nb = std::min<int64_t>(batch, max_it + 1 - enq);
    if (nb <= 0) {
      if ((rc = poll_drain(c, P, &done)) || done) break;
      nb = 1;  // not reached by construction; keeps the loop live if it ever is
    }
    for (int64_t q = 0; q < nb && !rc; ++q) {
      if ((rc = pa_cg_phase_a_t<T>(c, 2))) break;
      rc = pa_cg_phase_b_t<T>(c, 2);
      ++enq;
    }
    if (rc) break;
    cg_flush_fold<T>(c);
    rc = poll_submit(c, P, &done);
    batch = std::min<int64_t>(poll, std::max<int64_t>(1, enq));
  }
  if (!rc) rc = read_scalars(c);
  c->in_iterate = 0;
  if (rc) { c->fold_a_n = c->fold_b_n = c->fold_b_nsh = 0; return rc; }
  PA_HIP(c, hipEventRecord(c->ev1, c->stream));
  PA_HIP(c, hipEventSynchronize(c->ev1));
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
  fill_report(c, out, ms);
  c->solver_live = 0;
  return c->h_sc->err ? PA_E_NONFINITE : PA_OK;
}

template <typename T>
static int jacobi_run_t(pa_ctx* c, T* x, const T* rhs, double tol, int64_t max_it, double omega, pa_report* out) {
  const DevGeom& G = c->G;
  for (int q = 0; q < c->nterms; ++q)
    if (c->terms[q].kind != PA_OP_LAPLACIAN) { pa_set_err(c, "pa_jacobi: laplacian terms only"); return PA_E_ARG; }
  if (c->slab) { pa_set_err(c, "pa_jacobi is single-GPU only"); return PA_E_ARG; }
  const size_t fb = (size_t)G.ncell * sizeof(T);
  const int nblk = pa_grid_blocks(G.ncell);
  int rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_D0], &c->cap[SCR_D0], fb))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_PART], &c->cap[SCR_PART], (size_t)PA_MAX_PARTIALS * 4 * sizeof(double)))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_PART2], &c->cap[SCR_PART2], (size_t)3 * PA_MAX_GRID * sizeof(double)))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_SHELL], &c->cap[SCR_SHELL], 2 * (size_t)shell_elems(c) * sizeof(T)))) return rc;
  if ((rc = init_scalars(c, tol, max_it))) return rc;
  DevEq<T> E;
  pa_build_eq<T>(c, c->nterms, c->terms, E);
  const bool stat = bc_is_static(c);
  double* part = (double*)c->scr[SCR_PART];
  double* part2 = (double*)c->scr[SCR_PART2];
  // BC fill by the cheapest launch sequence, as in CG: closed form / one launch per axis / one per face
  c->bc_fused = bc_fusable(c);
  c->bc_pair = (!c->bc_fused && bc_pairable(c)) ? 1 : 0;
  c->shell_cur = 0;
  c->fold_a_n = c->fold_b_n = c->fold_b_nsh = 0;
  if (c->bc_fused) {
    if ((rc = bc_shell_fused<T>(c, x, nullptr, 0, false, nullptr, false))) return rc;
  } else if (c->bc_pair) {
    if ((rc = bc_pair_apply<T>(c, x, nullptr, 2, false, nullptr))) return rc;
  } else {
    if ((rc = bc_apply_t<T>(c, x))) return rc;
    hipLaunchKernelGGL(k_shell<T>, dim3(shell_blocks(c)), dim3(PA_BLOCK), 0, c->stream, G, c->sc, (const T*)x,
                       (T*)c->scr[SCR_SHELL], part2, 0);
  }
  T* buf[2] = {x, (T*)c->scr[SCR_D0]};
  int cur = 0;
  const int poll = poll_interval(c);
  PA_HIP(c, hipEventRecord(c->ev0, c->stream));
  int64_t enq = 0;
  // the stop test of sweep q is left to the prologue of sweep q+1 (pa_cg3d_kernel.h) when both are
  // tiled; this runs it as the single-block kernel it replaces (before a poll, before a generic sweep)
  auto flush = [&]() {
    if (c->fold_b_n > 0)
      hipLaunchKernelGGL(k_jacobi_post<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, c->fold_b_part, c->fold_b_n,
                         (const double*)part2, c->fold_b_nsh, pa_sums(c));
    c->fold_b_n = c->fold_b_nsh = 0;
  };
  PollPipe P;
  bool done = false;
  int64_t batch = 2;
  while (!done) {
    // the device stops by itself after max_it + 1 sweeps; sweeps are enqueued in pairs
    int64_t nb = std::min<int64_t>(batch, max_it + 2 - enq);
    if (nb <= 0) {
      if ((rc = poll_drain(c, P, &done))) return rc;
      if (done) break;
      nb = 2;
    }
    // two sweeps per round so that the iterate is back in the caller's buffer at every poll
    for (int64_t half = 0; half < ((nb + 1) & ~(int64_t)1); ++half) {
      Vec<T> xv = pa_vec_self<T>(c, buf[cur]);
      // partial rows alternate between the halves of SCR_PART: the next sweep reads these while it writes its own
      double* part_q = part + (cur ? 2 * (size_t)PA_MAX_PARTIALS : 0);
      if (c->profile) (void)hipEventRecord(c->pev[0], c->stream);   // slot 0: the sweep kernel
      int used = pa_tile3d_jacobi<T>(c, E, xv, rhs, buf[cur ^ 1], omega, part_q);
      if (used < 0) return used;
      const bool tiled = used > 0;
      if (!tiled) {
        flush();
        hipLaunchKernelGGL(k_jacobi<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, G, E, c->sc, xv, rhs, buf[cur ^ 1],
                           (T)omega, part_q);
        used = nblk;
      }
      if (c->profile) pa_profile_stop(c, 0);
      int nsh = 0;
      // NOTE: when done is set the sweep kernels return early, so buf[cur^1] is stale: the copy-back
      // below is guarded by the iteration parity recorded on the device (itr).
      if (!stat) {
        if (c->bc_fused) {
          if ((rc = bc_shell_fused<T>(c, buf[cur ^ 1], part2, 1, true, &nsh, false))) return rc;
        } else if (c->bc_pair) {
          if ((rc = bc_pair_apply<T>(c, buf[cur ^ 1], part2, 1, true, &nsh))) return rc;
        } else {
          if ((rc = bc_apply_t<T>(c, buf[cur ^ 1], true))) return rc;
          nsh = shell_blocks(c);
          hipLaunchKernelGGL(k_shell<T>, dim3(nsh), dim3(PA_BLOCK), 0, c->stream, G, c->sc, (const T*)buf[cur ^ 1],
                             (T*)c->scr[SCR_SHELL], part2, 1);
        }
      }
      if (c->fold && tiled && used <= PA_MAX_GRID && nsh <= 3 * PA_MAX_GRID) {
        c->fold_b_n = used;
        c->fold_b_nsh = nsh;
        c->fold_b_part = part_q;
      } else {
        hipLaunchKernelGGL(k_jacobi_post<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, part_q, used, part2, nsh,
                           pa_sums(c));
      }
      cur ^= 1;
      ++enq;
    }
    flush();
    if ((rc = poll_submit(c, P, &done))) return rc;
    batch = std::min<int64_t>(2 * poll, std::max<int64_t>(2, enq));
  }
  if ((rc = read_scalars(c))) return rc;
  // the final iterate lives in buf[itr & 1]
  if (c->h_sc->itr & 1) {
    hipLaunchKernelGGL(k_copy<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, (const T*)buf[1], x, G.ncell);
  }
  PA_HIP(c, hipEventRecord(c->ev1, c->stream));
  PA_HIP(c, hipEventSynchronize(c->ev1));
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
  fill_report(c, out, ms);
  return c->h_sc->err ? PA_E_NONFINITE : PA_OK;
}

template <typename T>
static int bicg_run_t(pa_ctx* c, T* x, const T* rhs, double tol, int64_t max_it, pa_report* out) {
  const DevGeom& G = c->G;
  if (c->slab) { pa_set_err(c, "pa_bicgstab is single-GPU only in this build"); return PA_E_ARG; }
  const size_t fb = (size_t)G.ncell * sizeof(T);
  const int nblk = pa_grid_blocks(G.ncell);
  int rc;
  const int ids[] = {SCR_R, SCR_D0, SCR_D1, SCR_R0, SCR_V0, SCR_V1, SCR_S, SCR_TT};
  for (int id : ids)
    if ((rc = pa_scratch(c, &c->scr[id], &c->cap[id], fb))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_PART], &c->cap[SCR_PART], (size_t)PA_MAX_PARTIALS * 6 * sizeof(double)))) return rc;
  if ((rc = init_scalars(c, tol, max_it))) return rc;
  DevEq<T> E;
  pa_build_eq<T>(c, c->nterms, c->terms, E);
  if ((rc = bc_apply_t<T>(c, x))) return rc;
  T* r = (T*)c->scr[SCR_R];
  T* r0 = (T*)c->scr[SCR_R0];
  T* p[2] = {(T*)c->scr[SCR_D0], (T*)c->scr[SCR_D1]};
  T* v[2] = {(T*)c->scr[SCR_V0], (T*)c->scr[SCR_V1]};
  T* s = (T*)c->scr[SCR_S];
  T* t = (T*)c->scr[SCR_TT];
  double* part = (double*)c->scr[SCR_PART];
  Vec<T> xv = pa_vec_self<T>(c, x);
  if ((rc = cg_residual_init<T>(c, E, xv, rhs, r0, r, (T*)nullptr, (T*)nullptr, part))) return rc;
  PA_HIP(c, hipMemsetAsync(p[0], 0, fb, c->stream));
  PA_HIP(c, hipMemsetAsync(v[0], 0, fb, c->stream));
  // rho_next = sum r0.r0 ; tol0 = sqrt(rho_next) ; first beta = rho_next / 1 * 1 / 1 (linalg.py:201-212)
  hipLaunchKernelGGL(k_cg_post_init<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, part, nblk, pa_sums(c), 2);
  if ((rc = read_scalars(c))) return rc;
  {
    SolverScalars h = *c->h_sc;
    h.rho_next = h.rr;
    h.tol = (double)(T)sqrt((T)h.rr);
    T b = (T)h.rho_next / (T)1.0;
    b = b * (T)1.0;
    b = b / (T)1.0;
    h.beta = (double)b;
    h.rho = h.rho_next;
    h.done = 0;  // `while not finished`: at least one iteration
    *c->h_sc = h;
    PA_HIP(c, hipMemcpyAsync(c->sc, c->h_sc, sizeof(h), hipMemcpyHostToDevice, c->stream));
    PA_HIP(c, hipStreamSynchronize(c->stream));
  }
  int cur = 0;
  const int poll = poll_interval(c);
  PA_HIP(c, hipEventRecord(c->ev0, c->stream));
  int64_t enq = 0;
  // The three single-block scalar kernels of an iteration are folded into the prologue of the kernel
  // that follows each (pa_cg3d_kernel.h phases 5 / 6, k_bicg_x) when that kernel is a tiled one / the
  // row counts are small; each producer has its own region of SCR_PART, because its consumer reads the
  // rows while writing its own.  `pend*` = rows waiting for a prologue.
  double* const reg0 = part;                                   // r0.v'            (1 column)
  double* const reg1 = part + (size_t)PA_MAX_PARTIALS;         // |s|^2 t.s t.t r0.t (4 columns)
  double* const reg2 = part + 5 * (size_t)PA_MAX_PARTIALS;     // |r|^2            (1 column)
  const bool fold = c->fold && !c->slab;
  int pend3 = 0;
  c->fold_a_n = c->fold_b_n = c->fold_b_nsh = 0;
  auto flush3 = [&]() {
    if (pend3 > 0) hipLaunchKernelGGL(k_bicg_post<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, reg2, pend3, 3);
    pend3 = 0;
    c->fold_b_n = 0;
  };
  PollPipe P;
  bool done = false;
  int64_t batch = 1;
  const int64_t max_enq = std::max<int64_t>(max_it, 1);  // the device stops by itself after max_it iterations
  while (!done) {
    int64_t nb = std::min<int64_t>(batch, max_enq - enq);
    if (nb <= 0) {
      if ((rc = poll_drain(c, P, &done))) return rc;
      if (done) break;
      nb = 1;
    }
   for (int64_t qi = 0; qi < nb; ++qi) {
    Vec<T> rv = pa_vec_self<T>(c, r), pv = pa_vec_self<T>(c, p[cur]), vv = pa_vec_self<T>(c, v[cur]);
    c->fold_b_n = pend3;          // phase 5 closes the previous iteration (and swaps the scalar slots)
    c->fold_b_part = reg2;
    int used = pa_tile3d_bicg_pv<T>(c, E, rv, pv, vv, (const T*)r0, p[cur ^ 1], v[cur ^ 1], reg0);
    if (used < 0) return used;
    if (used > 0) {
      pend3 = 0;
    } else {
      flush3();
      hipLaunchKernelGGL(k_bicg_pv<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, G, E, c->sc, rv, pv, vv, (const T*)r0,
                         p[cur ^ 1], v[cur ^ 1], reg0);
      used = nblk;
    }
    int pend0 = (fold && used <= PA_MAX_GRID) ? used : 0;
    if (!pend0) hipLaunchKernelGGL(k_bicg_post<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, reg0, used, 0);
    Vec<T> vnv = pa_vec_self<T>(c, v[cur ^ 1]);
    c->fold_a_n = pend0;          // phase 6 computes alpha itself
    int used2 = pa_tile3d_bicg_st<T>(c, E, rv, vnv, (const T*)r0, s, t, reg1);
    c->fold_a_n = 0;
    if (used2 < 0) return used2;
    int pend12 = 0;
    if (used2 > 0) {
      pend12 = (fold && used2 <= PA_MAX_GRID) ? used2 : 0;
      if (!pend12) hipLaunchKernelGGL(k_bicg_post<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, reg1, used2, 12);
    } else {
      if (pend0) hipLaunchKernelGGL(k_bicg_post<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, reg0, pend0, 0);
      hipLaunchKernelGGL(k_bicg_s<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, G, c->sc, (const T*)r,
                         (const T*)v[cur ^ 1], s, reg1);
      hipLaunchKernelGGL(k_bicg_post<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, reg1, nblk, 1);
      Vec<T> sv = pa_vec_self<T>(c, s);
      hipLaunchKernelGGL(k_bicg_t<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, G, E, c->sc, sv, (const T*)r0, t, reg1);
      hipLaunchKernelGGL(k_bicg_post<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, reg1, nblk, 2);
    }
    hipLaunchKernelGGL(k_bicg_x<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, G, c->sc, x, (const T*)p[cur ^ 1],
                       (const T*)s, (const T*)t, r, reg2, (const double*)reg1, pend12, c->sc);
    if ((rc = bc_apply_auto<T>(c, x, true))) return rc;
    if (fold && nblk <= PA_MAX_GRID)
      pend3 = nblk;
    else
      hipLaunchKernelGGL(k_bicg_post<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, reg2, nblk, 3);
    cur ^= 1;
    ++enq;
   }
    flush3();
    if ((rc = poll_submit(c, P, &done))) return rc;
    batch = std::min<int64_t>(poll, std::max<int64_t>(1, enq));
  }
  if ((rc = read_scalars(c))) return rc;
  PA_HIP(c, hipEventRecord(c->ev1, c->stream));
  PA_HIP(c, hipEventSynchronize(c->ev1));
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
  fill_report(c, out, ms);
  return c->h_sc->err ? PA_E_NONFINITE : PA_OK;
}

extern "C" {

int pa_cg(pa_ctx* c, void* x, const void* rhs, double tol, int64_t max_it, pa_report* out) {
  if (!c || !c->grid_set || !c->eq_set) { if (c) pa_set_err(c, "pa_cg: grid/equation not set"); return PA_E_STATE; }
  if (int rc0 = check_eq_applicable(c)) return rc0;
  if (!out) return PA_E_ARG;
  if (c->slab) { pa_set_err(c, "pa_cg is the single-GPU loop; use the stepwise API on a slab"); return PA_E_STATE; }
  PA_HIP(c, hipSetDevice(c->device));
  return c->dtype == PA_F64 ? cg_run_t<double>(c, (double*)x, (const double*)rhs, tol, max_it, out)
                            : cg_run_t<float>(c, (float*)x, (const float*)rhs, tol, max_it, out);
}

int pa_bicgstab(pa_ctx* c, void* x, const void* rhs, double tol, int64_t max_it, pa_report* out) {
  if (!c || !c->grid_set || !c->eq_set) { if (c) pa_set_err(c, "pa_bicgstab: grid/equation not set"); return PA_E_STATE; }
  if (int rc0 = check_eq_applicable(c)) return rc0;
  if (!out) return PA_E_ARG;
  PA_HIP(c, hipSetDevice(c->device));
  return c->dtype == PA_F64 ? bicg_run_t<double>(c, (double*)x, (const double*)rhs, tol, max_it, out)
                            : bicg_run_t<float>(c, (float*)x, (const float*)rhs, tol, max_it, out);
}

int pa_jacobi(pa_ctx* c, void* x, const void* rhs, double tol, int64_t max_it, double omega, pa_report* out) {
  if (!c || !c->grid_set || !c->eq_set) { if (c) pa_set_err(c, "pa_jacobi: grid/equation not set"); return PA_E_STATE; }
  if (int rc0 = check_eq_applicable(c)) return rc0;
  if (!out) return PA_E_ARG;
  PA_HIP(c, hipSetDevice(c->device));
  return c->dtype == PA_F64 ? jacobi_run_t<double>(c, (double*)x, (const double*)rhs, tol, max_it, omega, out)
                            : jacobi_run_t<float>(c, (float*)x, (const float*)rhs, tol, max_it, omega, out);
}

}  // extern "C"

// ============================================================================
//  stepwise CG (bench.py, slab-decomposed driver)
// ============================================================================
extern "C" {

int pa_cg_begin(pa_ctx* c, void* x, const void* rhs, double tol, int64_t max_it) {
  if (!c || !c->grid_set || !c->eq_set) { if (c) pa_set_err(c, "pa_cg_begin: grid/equation not set"); return PA_E_STATE; }
  if (int rc0 = check_eq_applicable(c)) return rc0;
  PA_HIP(c, hipSetDevice(c->device));
  return c->dtype == PA_F64 ? cg_begin_t<double>(c, (double*)x, (const double*)rhs, tol, max_it)
                            : cg_begin_t<float>(c, (float*)x, (const float*)rhs, tol, max_it);
}

int pa_cg_phase_a(pa_ctx* c) {
  if (!c || !c->solver_live) { if (c) pa_set_err(c, "pa_cg_phase_a without pa_cg_begin"); return PA_E_STATE; }
  const int st = c->slab ? 0 : 2;
  return c->dtype == PA_F64 ? pa_cg_phase_a_t<double>(c, st) : pa_cg_phase_a_t<float>(c, st);
}

int pa_cg_phase_b(pa_ctx* c) {
  if (!c || !c->solver_live) { if (c) pa_set_err(c, "pa_cg_phase_b without pa_cg_begin"); return PA_E_STATE; }
  if (c->slab) {  // alpha from the all-reduced sum d.Ad
    if (c->dtype == PA_F64)
      hipLaunchKernelGGL(k_cg_post_a<double>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, (const double*)nullptr, 0, pa_sums(c), 1);
    else
      hipLaunchKernelGGL(k_cg_post_a<float>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, (const double*)nullptr, 0, pa_sums(c), 1);
  }
  const int st = c->slab ? 0 : 2;
  return c->dtype == PA_F64 ? pa_cg_phase_b_t<double>(c, st) : pa_cg_phase_b_t<float>(c, st);
}

int pa_cg_bc(pa_ctx* c) {
  if (!c || !c->solver_live) { if (c) pa_set_err(c, "pa_cg_bc without pa_cg_begin"); return PA_E_STATE; }
  if (!c->slab) return PA_OK;  // done inside phase_b
  return c->dtype == PA_F64 ? pa_cg_bc_t<double>(c) : pa_cg_bc_t<float>(c);
}

int pa_cg_finish_iter(pa_ctx* c) {
  if (!c || !c->solver_live) { if (c) pa_set_err(c, "pa_cg_finish_iter without pa_cg_begin"); return PA_E_STATE; }
  if (!c->slab) return PA_OK;  // logic already ran inside phase_b
  if (c->dtype == PA_F64)
    hipLaunchKernelGGL(k_cg_post_b<double>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, (const double*)nullptr, 0,
                       (const double*)nullptr, 0, pa_sums(c), 1);
  else
    hipLaunchKernelGGL(k_cg_post_b<float>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, (const double*)nullptr, 0,
                       (const double*)nullptr, 0, pa_sums(c), 1);
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

static int cg_one_iteration(pa_ctx* c) {
  int rc = c->dtype == PA_F64 ? pa_cg_phase_a_t<double>(c, 2) : pa_cg_phase_a_t<float>(c, 2);
  if (rc) return rc;
  return c->dtype == PA_F64 ? pa_cg_phase_b_t<double>(c, 2) : pa_cg_phase_b_t<float>(c, 2);
}

// PYAPES_HIP_GRAPH=1: replay a captured pair of iterations (the direction buffers ping-pong, so the
// launch sequence has period 2) as a hipGraph instead of enqueueing every kernel again
static int cg_iterate_graph(pa_ctx* c, int64_t n, int64_t* done) {
  *done = 0;
  if (n < 4 || c->profile || c->overlap) return PA_OK;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return PA_OK; }
  int rc = cg_one_iteration(c);
  if (!rc) rc = cg_one_iteration(c);
  hipError_t e = hipStreamEndCapture(c->stream, &graph);
  if (rc || e != hipSuccess || !graph) { (void)hipGetLastError(); if (graph) (void)hipGraphDestroy(graph); return rc ? rc : PA_OK; }
  if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipGraphDestroy(graph); return PA_OK; }
  // the capture itself executed nothing: all n iterations are still to do
  const int64_t pairs = n / 2;
  for (int64_t q = 0; q < pairs; ++q)
    if (hipGraphLaunch(exec, c->stream) != hipSuccess) { rc = pa_hip_fail(c, hipGetLastError(), "hipGraphLaunch"); break; }
  (void)hipGraphExecDestroy(exec);
  (void)hipGraphDestroy(graph);
  if (!rc) *done = 2 * pairs;
  return rc;
}

int pa_cg_iterate(pa_ctx* c, int64_t n) {
  if (!c || !c->solver_live) { if (c) pa_set_err(c, "pa_cg_iterate without pa_cg_begin"); return PA_E_STATE; }
  if (c->slab) { pa_set_err(c, "pa_cg_iterate is single-rank; drive the phases on a slab"); return PA_E_STATE; }
  static int use_graph = -1;
  if (use_graph < 0) { const char* g = getenv("PYAPES_HIP_GRAPH"); use_graph = (g && atoi(g) != 0) ? 1 : 0; }
  if (use_graph) {
    int64_t done = 0;
    if (int rc = cg_iterate_graph(c, n, &done)) return rc;
    n -= done;
  }
  c->in_iterate = use_graph ? 0 : 1;
  int rc = PA_OK;
  for (int64_t q = 0; q < n && !rc; ++q) {
    rc = c->dtype == PA_F64 ? pa_cg_phase_a_t<double>(c, 2) : pa_cg_phase_a_t<float>(c, 2);
    if (!rc) rc = c->dtype == PA_F64 ? pa_cg_phase_b_t<double>(c, 2) : pa_cg_phase_b_t<float>(c, 2);
  }
  c->in_iterate = 0;
  if (c->dtype == PA_F64) cg_flush_fold<double>(c); else cg_flush_fold<float>(c);  // the last iteration's stop test
  if (rc) return rc;
  return pa_join_side(c);  // the ctx stream now covers everything that was enqueued
}

int pa_profile_set(pa_ctx* c, int on) {
  if (!c) return PA_E_ARG;
  if (on && !c->pev[0])
    for (int q = 0; q < 4; ++q) PA_HIP(c, hipEventCreate(&c->pev[q]));
  c->profile = on ? 1 : 0;
  c->prof_ms[0] = c->prof_ms[1] = 0.0;
  c->prof_n[0] = c->prof_n[1] = 0;
  return PA_OK;
}

int pa_profile_read(pa_ctx* c, double* ms_a, int64_t* n_a, double* ms_b, int64_t* n_b) {
  if (!c) return PA_E_ARG;
  if (ms_a) *ms_a = c->prof_ms[0];
  if (n_a) *n_a = c->prof_n[0];
  if (ms_b) *ms_b = c->prof_ms[1];
  if (n_b) *n_b = c->prof_n[1];
  return PA_OK;
}

int pa_report_read(pa_ctx* c, pa_report* out) {
  if (!c || !out) return PA_E_ARG;
  int rc = read_scalars(c);
  if (rc) return rc;
  fill_report(c, out, 0.f);
  return PA_OK;
}

int pa_cg_end(pa_ctx* c, pa_report* out) {
  if (!c || !c->solver_live) { if (c) pa_set_err(c, "pa_cg_end without pa_cg_begin"); return PA_E_STATE; }
  int rc = out ? pa_report_read(c, out) : PA_OK;
  c->solver_live = 0;
  if (rc) return rc;
  return (out && out->status) ? PA_E_NONFINITE : PA_OK;
}

}  // extern "C"
