// pa_bc.hip -- the ordered BC fill (pyapes/variables/bcs.py:200-280, linalg.py:282-299) and the
// boundary-shell term of the stop test: one launch per face (any order), one per axis (factory order),
// or the closed form (small shells).  All three bit-identical (tests/test_gpu_bc_fused.py, _pair.py).
#include "pa_host.h"

#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>

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

// A kernel argument struct that is indexed with a run-time index (E.t[q], B.f[face]) gets copied to
// scratch memory by the compiler, and a kernel with a private segment costs ~10 us more per dispatch
// on MI355X (measured: 64^3 CG iteration 29 -> 52 us).  Reading the struct in place -- through the
// kernarg segment pointer -- keeps such kernels off scratch.  `off` = byte offset of the parameter.
template <typename S>
__device__ __forceinline__ const S& pa_kernarg(size_t off) {
  return *(const S*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + off);
}
static_assert(sizeof(DevGeom) % 8 == 0, "second kernel parameter starts at sizeof(DevGeom)");

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
// PER = false: no periodic face anywhere (the one-pass form of pa_bc_shell_fused): the periodic branches of the
// three stages -- three more recursive evaluations each -- are compiled out (18,000 -> a third of the
// instructions; the kernel is one short thread per shell node and instruction-bound)
template <typename T, bool PER = true>
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
    if (PER && B.slab_periodic0 && B.f[f].type == 4) {
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
    if (type == 3 || !PER) return raw0(f, lower ? 1 : N - 2, base);
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
    if (type == 3 || !PER) return stage0(i, lower ? 1 : N - 2, k);
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
    if (type == 3 || !PER) return stage1(i, j, lower ? 1 : N - 2);
    T t1 = stage1(i, j, 1) - stage1(i, j, N - 1);
    return t1 + stage1(i, j, N - 2);
  }
};

template <typename T, bool PER>
__device__ __forceinline__ T pa_bc_v3(const DevGeom& G, const BCAll<T>& B, const T* __restrict__ x, int64_t i,
                                      int64_t j, int64_t k) {
  BCEval<T, PER> ev{G, B, x};
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
  // local = u * n + v: a 32-bit division where the mesh allows it (a quarter of the 64-bit one; the shell
  // kernels are instruction-bound: one short thread per node)
  const int64_t dn = a == 2 ? G.n1 : G.n2;
  int64_t u, v;
  if (G.ncell <= 0x7fffffffLL) {
    const uint32_t lu = (uint32_t)local, du = (uint32_t)dn, uu = lu / du;
    u = uu; v = lu - uu * du;
  } else {
    u = local / dn; v = local - u * dn;
  }
  if (a == 0) {
    int64_t gi = side == 0 ? 0 : G.g0 - 1;
    i = gi - G.off0;
    if (i < 0 || i >= G.n0) return false;
    j = u; k = v;
  } else if (a == 1) {
    i = u; k = v;
    j = side == 0 ? 0 : G.n1 - 1;
    int64_t gi = i + G.off0;
    if (G.act[0] && (gi == 0 || gi == G.g0 - 1)) return false;
  } else {
    i = u; j = v;
    k = side == 0 ? 0 : G.n2 - 1;
    int64_t gi = i + G.off0;
    if (G.act[0] && (gi == 0 || gi == G.g0 - 1)) return false;
    if (G.act[1] && (j == 0 || j == G.n1 - 1)) return false;
  }
  return true;
}

template <typename T, bool PER>
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
    T v = pa_bc_v3<T, PER>(G, B, x, i, j, k);
#ifdef PA_DEBUG_BC
    if (i == 0 && j == 1 && k == G.n2 - 1) {
      BCEval<T, PER> ev{G, B, x};
      printf("DBG node(0,1,%lld) v=%g stage1(0,1,n2-2)=%g stage0=%g types %d %d %d %d %d %d slabp %d far %p %p %p raw1 %g rawN1 %g rawN2 %g\n",
             (long long)k, (double)v, (double)ev.stage1(0, 1, G.n2 - 2), (double)ev.stage0(0, 1, G.n2 - 2),
             B.f[0].type, B.f[1].type, B.f[2].type, B.f[3].type, B.f[4].type, B.f[5].type, B.slab_periodic0,
             (void*)B.far_lo0, (void*)B.far_lo1, (void*)B.far_hi0, (double)ev.raw0(0, 1, 1 * G.s1 + G.n2 - 2),
             (double)ev.raw0(0, G.g0 - 1, 1 * G.s1 + G.n2 - 2), (double)ev.raw0(0, G.g0 - 2, 1 * G.s1 + G.n2 - 2));
    }
#endif
    shell_new[q] = v;
    if (xw) xw[i * G.s0 + j * G.s1 + k] = v;  // single pass (see pa_bc_shell_fused): no later read sees this node
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
                                                       double* __restrict__ partials, int mode) {
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
  if (mode == 1) pa_block_reduce_store<1>(s, partials);
}

// ---- host side ----------------------------------------------------------------------------------
template <typename T>
int pa_bc_apply_faces(pa_ctx* c, T* x, bool guarded) {
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
    hipLaunchKernelGGL(k_bc_face<T>, dim3(pa_grid_blocks(nu * nv)), dim3(PA_BLOCK), 0, c->stream, G, x, B);
  }
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

// fewest launches that keep the sequential semantics: closed form (2) for small shells, one launch per
// axis (<= 3) for the factory order, else one per face in list order
template <typename T>
int pa_bc_apply_auto(pa_ctx* c, T* x, bool guarded) {
  if (pa_bc_fusable(c)) return pa_bc_shell_fused<T>(c, x, nullptr, 0, guarded, nullptr, true);
  if (pa_bc_pairable(c)) return pa_bc_pair_apply<T>(c, x, nullptr, 0, guarded, nullptr);
  return pa_bc_apply_faces<T>(c, x, guarded);
}

int pa_bc_apply_any(pa_ctx* c, void* x) {
  return c->dtype == PA_F64 ? pa_bc_apply_auto<double>(c, (double*)x, false) : pa_bc_apply_auto<float>(c, (float*)x, false);
}

extern "C" {

int pa_apply_bc(pa_ctx* c, void* x) {
  if (!c || !c->grid_set) return PA_E_STATE;
  PA_HIP(c, hipSetDevice(c->device));
  return pa_bc_apply_any(c, x);
}

}  // extern "C"

int pa_shell_blocks(const pa_ctx* c) {
  const DevGeom& G = c->G;
  int64_t tot = 0;
  if (G.act[0]) tot += 2 * G.n1 * G.n2;
  if (G.act[1]) tot += 2 * G.n0 * G.n2;
  if (G.act[2]) tot += 2 * G.n0 * G.n1;
  return pa_grid_blocks(tot);
}
int64_t pa_shell_elems(const pa_ctx* c) {
  const DevGeom& G = c->G;
  return 2 * (G.n1 * G.n2 + G.n0 * G.n2 + G.n0 * G.n1);
}

// B(x) is a no-op after the first fill when every face is dirichlet (values frozen per solve)
bool pa_bc_is_static(const pa_ctx* c) {
  for (int f = 0; f < 6; ++f) {
    int t = c->bc[f].type;
    if (t != PA_BC_NONE && t != PA_BC_DIRICHLET) return false;
  }
  return true;
}


// ---- fused BC fill (+ shell stop-test term) ------------------------------------------------------
// usable when the faces are listed in the factory order and every mesh axis has >= 5 nodes
bool pa_bc_fusable(const pa_ctx* c) {
  if (c->bc_path & 1) return false;   // option "bc_path" (tests: the paths are bit-identical)
  // Measured on MI355X (512^3 fp64 periodic): the closed form costs 86 + 40 us against 62 + 19 us for
  // six face launches + the shell pass, so it only wins where launches, not bytes, set the time.
  // Against the per-axis pair kernels (explicit Euler step, fp32, us / step fused : pair : faces):
  // 64^3 17 : 20 : 25, 128^3 27.9 : 28.5 : 33, 192^3 44 : 40 : 45, 256^3 65 : 55 : 59 -- the crossover
  // sits between 98 k and 221 k shell nodes.  Option "bc_path" bit 2 forces the closed form (tests do).
  // Round 2: without a periodic face (and off a slab) the closed form is ONE launch writing straight into the
  // field (pa_bc_shell_fused), and the explicit Euler step in front of it got faster, so the crossover moved:
  // Euler step fp32, us / step fused : pair -- 128^3 17.2 : 23.6, 256^3 35.6 : 40.4, 384^3 134 : 140,
  // 512^3 253 : 258; CG fp32 mixed faces 128^3 43.7 : 48.2, 256^3 132 : 133, 384^3 417 : 417.
  bool one_pass = !c->slab;
  for (int f = 0; f < 6; ++f)
    if (c->bc[f].type == PA_BC_PERIODIC) one_pass = false;
  const int64_t limit = pa_bc_pairable(c) ? (one_pass ? 2000000 : 150000) : 400000;
  if (!(c->bc_path & 4) &&
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
int pa_bc_shell_fused(pa_ctx* c, T* x, double* part2, int with_delta, bool guarded, int* nsh,
                          bool standalone) {
  BCAll<T> B;
  int rc = bc_fill_all<T>(c, B);
  if (rc) return rc;
  const int nb = pa_shell_blocks(c);
  const size_t half = (size_t)pa_shell_elems(c);
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
  bool direct = !c->slab;
  for (int f = 0; f < 6; ++f)
    if (c->bc[f].type == PA_BC_PERIODIC) direct = false;
  bool any_per = false;
  for (int f = 0; f < 6; ++f)
    if (c->bc[f].type == PA_BC_PERIODIC) any_per = true;
  if (any_per)
    hipLaunchKernelGGL((k_bc_compute<T, true>), dim3(nb), dim3(PA_BLOCK), 0, c->stream, c->G, B, done, (const T*)x,
                       (const T*)so, sn, part2, with_delta, direct ? x : (T*)nullptr);
  else
    hipLaunchKernelGGL((k_bc_compute<T, false>), dim3(nb), dim3(PA_BLOCK), 0, c->stream, c->G, B, done, (const T*)x,
                       (const T*)so, sn, part2, with_delta, direct ? x : (T*)nullptr);
  if (!direct)
    hipLaunchKernelGGL(k_bc_scatter<T>, dim3(nb), dim3(PA_BLOCK), 0, c->stream, c->G, done, x, (const T*)sn);
  if (!standalone) c->shell_cur ^= 1;
  if (nsh) *nsh = nb;
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

// BC list in factory order with both faces of every mesh axis present (what the BC factories emit),
// >= 5 nodes per axis: the per-axis pair kernels apply
bool pa_bc_pairable(const pa_ctx* c) {
  if (c->bc_path & 2) return false;
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
int pa_bc_pair_apply(pa_ctx* c, T* x, double* part2, int mode, bool guarded, int* nsh) {
  const DevGeom& G = c->G;
  const int64_t sz[3] = {G.n1 * G.n2, G.n0 * G.n2, G.n0 * G.n1};
  int64_t start[6], total = 0;
  for (int f = 0; f < 6; ++f) { start[f] = total; total += G.act[f >> 1] ? sz[f >> 1] : 0; }
  T* shell = (T*)c->scr[SCR_SHELL];
  int rows = 0;
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
    hipLaunchKernelGGL(k_bc_pair<T>, dim3(nb), dim3(PA_BLOCK), 0, c->stream, G, x, P,
                       guarded ? pa_done_flag(c) : (const int*)nullptr, shell, part2 ? part2 + rows : nullptr,
                       mode);
    if (mode == 1) rows += nb;
  }
  if (nsh) *nsh = rows;
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}


// partial rows the BC fill + boundary-shell pass of one solver iteration writes, by the path pa_cg_begin chose
int pa_bc_shell_rows(const pa_ctx* c) {
  if (c->bc_fused || !c->bc_pair) return pa_shell_blocks(c);
  const DevGeom& G = c->G;
  const int64_t sz[3] = {G.n1 * G.n2, G.n0 * G.n2, G.n0 * G.n1};
  int rows = 0;
  for (int a = 0; a < 3; ++a) {
    if (!G.act[a]) continue;
    int lo = c->bc[2 * a].type, hi = c->bc[2 * a + 1].type;
    if (a == 0) {  // slab: only the rank holding the global boundary plane applies the face
      if (G.off0 != 0) lo = 0;
      if (G.off0 + G.n0 != G.g0) hi = 0;
    }
    if (lo || hi) rows += pa_grid_blocks(sz[a]);
  }
  return rows;
}

// boundary-shell pass on its own (after a face-by-face fill): save the shell, with_delta: + partial sums of
// (new - old)^2 -> part2 (pa_shell_blocks rows)
template <typename T>
void pa_shell_launch(pa_ctx* c, const T* x, T* shell, double* part2, int with_delta) {
  hipLaunchKernelGGL(k_shell<T>, dim3(pa_shell_blocks(c)), dim3(PA_BLOCK), 0, c->stream, c->G, c->sc, x, shell, part2,
                     with_delta);
}

#define PA_BC_INST(T)                                                                            \
  template int pa_bc_apply_faces<T>(pa_ctx*, T*, bool);                                          \
  template int pa_bc_apply_auto<T>(pa_ctx*, T*, bool);                                           \
  template int pa_bc_shell_fused<T>(pa_ctx*, T*, double*, int, bool, int*, bool);                \
  template int pa_bc_pair_apply<T>(pa_ctx*, T*, double*, int, bool, int*);                       \
  template void pa_shell_launch<T>(pa_ctx*, const T*, T*, double*, int);
PA_BC_INST(float)
PA_BC_INST(double)
#undef PA_BC_INST
