// pa_device.h -- device-side data model and the literal stencil evaluation shared
// by every kernel of libpyapes_hip (gfx950 only).
//
// Arithmetic contract (parity with the reference torch-CPU path, bit-exact for
// operator / BC outputs): every product and sum below is a separately rounded
// IEEE operation in the field dtype, in the reference's order
//   per axis  ((cP*x+) + cC*x) + cM*x-            pyapes/solver/fdc.py:190-198
//   axes 0->1->2 accumulated into zero            fdc.py:103-108
//   * param, * sign, summed over terms in order   fdm.py:166-169, ops.py:140-149
// The translation unit is compiled with -ffp-contract=off so that hipcc never
// fuses them into FMAs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PA_MAX_TERMS 4
#define PA_BLOCK 256

// internal axes: always 3, fastest = 2.  A d-dimensional user mesh occupies the
// LAST d internal axes (1-D -> (1,1,n0), 2-D -> (1,n0,n1)), so the contiguous
// axis is always internal axis 2 and the slab axis of a 3-D mesh is internal 0.
struct DevGeom {
  int64_t n0, n1, n2;   // local extents
  int64_t s0, s1;       // element strides of axes 0,1 (axis 2 stride = 1)
  int64_t g0, off0;     // global extent of axis 0 and global index of local plane 0
  int64_t ncell;        // n0*n1*n2
  int act[3];           // axis is a mesh axis (extent > 1 and mapped)
  int64_t slo[3], shi[3]; // interior set S in GLOBAL indices, inclusive (mesh/tools.py:7-20)
  int bct[6];           // BC type per internal face 2*axis+side
  int treat[6];         // neumann|symmetry (bcs.py:158-163)
  int hi_last[3];       // upper face of the axis is listed after its lower face
};

template <typename T>
struct LapCoef {  // fdc.py:376-423
  T inv[3];    // fl(1 / fl(h*h))
  T m2inv[3];  // fl(-2 / fl(h*h))
  T c23[3];    // fl(fl(2/3) / fl(h*h))
};

template <typename T>
struct GradCoef {  // fdc.py:543-609 with gamma = 1
  T g[3];      // fl( 1 / fl(2h))
  T mg[3];     // fl(-1 / fl(2h))
  T lo_p[3];   // fl(fl(1 + 1/3) / 2h)   lower neumann/symmetry row: Ap
  T lo_c[3];   // fl(fl(0 - 1/3) / 2h)                                Ac
  T hi_c[3];   // fl(fl(0 + 1/3) / 2h)   upper row: Ac
  T hi_m[3];   // fl(fl(-1 - 1/3) / 2h)             Am
  T h2[3];     // fl(2h)
  T h[3];      // h
  T ih[3];     // fl(1 / h)
};

template <typename T>
struct DevTerm {
  int kind;
  int has_coeff;
  T sign;
  T coeff;
  const T* coeff_f;
  T u;
  const T* u_f;
};

template <typename T>
struct DevEq {
  int nterms;
  DevTerm<T> t[PA_MAX_TERMS];
  LapCoef<T> lap;
  GradCoef<T> grd;
  // axisymmetric (rz) mesh: 5 x rz_n table of pa_coord_set, r = internal axis 1; null on xyz meshes
  const T* rz;
  int64_t rz_n;
};

#define PA_RZ_AXIS 1  // 2-D meshes are right-aligned in the 3 internal axes: r -> 1, z -> 2

// a field plus the two ghost planes that complete it along internal axis 0
// (slab neighbours, or the wrap-around planes of the field itself when P = 1)
template <typename T>
struct Vec {
  const T* p;
  const T* glo;  // plane "-1"
  const T* ghi;  // plane "n0"
};

__device__ __forceinline__ void pa_decode(const DevGeom& G, int64_t idx, int64_t& i, int64_t& j, int64_t& k) {
  if (G.ncell <= 0x7fffffffLL) {  // uniform: 32-bit divisions cost a quarter of the 64-bit ones
    const uint32_t id = (uint32_t)idx, s0 = (uint32_t)G.s0, s1 = (uint32_t)G.s1;
    const uint32_t ii = id / s0, rem = id - ii * s0, jj = rem / s1;
    i = ii;
    j = jj;
    k = rem - jj * s1;
    return;
  }
  i = idx / G.s0;
  int64_t rem = idx - i * G.s0;
  j = rem / G.s1;
  k = rem - j * G.s1;
}

__device__ __forceinline__ int64_t pa_wrap(int64_t v, int64_t n) {
  return v < 0 ? v + n : (v >= n ? v - n : v);
}

// plain field access
template <typename T>
struct FieldAcc {
  Vec<T> v;
  __device__ __forceinline__ T at(const DevGeom& G, int64_t i, int64_t j, int64_t k) const {
    if (i < 0) return v.glo[j * G.s1 + k];
    if (i >= G.n0) return v.ghi[j * G.s1 + k];
    return v.p[i * G.s0 + j * G.s1 + k];
  }
};

// the CG search direction before it is stored: d' = r + beta d (linalg.py:141)
template <typename T>
struct DirAcc {
  Vec<T> r, d;
  T beta;
  __device__ __forceinline__ T at(const DevGeom& G, int64_t i, int64_t j, int64_t k) const {
    // pointers first, ONE load per field after: with the loads inside the three branches the compiler
    // sinks them behind a phi of member addresses, i.e. an indexed read of a stack copy of *this, and a
    // kernel with a private segment costs ~10 us more to dispatch
    const int64_t o = j * G.s1 + k;
    const T* rb = r.p + i * G.s0;
    const T* db = d.p + i * G.s0;
    if (i < 0) { rb = r.glo; db = d.glo; }
    if (i >= G.n0) { rb = r.ghi; db = d.ghi; }
    const T rv = rb[o], dv = db[o];
    T bd = beta * dv;
    return rv + bd;
  }
};

template <typename T, class Acc>
__device__ __forceinline__ void pa_nbrs(const DevGeom& G, const Acc& a, int ax, int64_t i, int64_t j,
                                        int64_t k, T& xp, T& xm) {
  if (ax == 0) {
    xp = a.at(G, i + 1, j, k);
    xm = a.at(G, i - 1, j, k);
  } else if (ax == 1) {
    xp = a.at(G, i, pa_wrap(j + 1, G.n1), k);
    xm = a.at(G, i, pa_wrap(j - 1, G.n1), k);
  } else {
    xp = a.at(G, i, j, pa_wrap(k + 1, G.n2));
    xm = a.at(G, i, j, pa_wrap(k - 1, G.n2));
  }
}

__device__ __forceinline__ void pa_gidx(const DevGeom& G, int64_t i, int64_t j, int64_t k, int64_t* g,
                                        int64_t* N) {
  g[0] = i + G.off0; g[1] = j; g[2] = k;
  N[0] = G.g0; N[1] = G.n1; N[2] = G.n2;
}

__device__ __forceinline__ bool pa_in_S(const DevGeom& G, int64_t i, int64_t j, int64_t k) {
  int64_t gi = i + G.off0;
  return gi >= G.slo[0] && gi <= G.shi[0] && j >= G.slo[1] && j <= G.shi[1] && k >= G.slo[2] &&
         k <= G.shi[2];
}

__device__ __forceinline__ bool pa_on_shell(const DevGeom& G, int64_t i, int64_t j, int64_t k) {
  int64_t gi = i + G.off0;
  return (G.act[0] && (gi == 0 || gi == G.g0 - 1)) || (G.act[1] && (j == 0 || j == G.n1 - 1)) ||
         (G.act[2] && (k == 0 || k == G.n2 - 1));
}

// which boundary-row special case applies on axis ax at global index g (0 none, 1 lower, 2 upper)
__device__ __forceinline__ int pa_row_case(const DevGeom& G, int ax, int64_t g, int64_t N, const int* flag) {
  bool lo = flag[2 * ax] && g == 1;
  bool hi = flag[2 * ax + 1] && g == N - 2;
  if (lo && hi) {
    if (G.hi_last[ax]) lo = false; else hi = false;
  }
  return lo ? 1 : (hi ? 2 : 0);
}

// sum_k sign_k * param_k * Op_k(acc)   at local node (i,j,k); xc = acc.at(i,j,k)
template <typename T, class Acc>
__device__ __forceinline__ T pa_apply_terms(const DevGeom& G, const DevEq<T>& E, const Acc& acc, int64_t i,
                                            int64_t j, int64_t k, T xc) {
  int64_t g[3], N[3];
  pa_gidx(G, i, j, k, g, N);
  const int64_t o = i * G.s0 + j * G.s1 + k;
  T res = (T)0;
#pragma unroll 1
  for (int q = 0; q < E.nterms; ++q) {
    const DevTerm<T>& t = E.t[q];
    T ax = (T)0;
    if (t.kind == 0) {  // PA_OP_LAPLACIAN
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        if (!G.act[a]) continue;
        T cP = E.lap.inv[a], cC = E.lap.m2inv[a], cM = E.lap.inv[a];
        T cB = E.lap.c23[a];
        if (E.rz && a == PA_RZ_AXIS) {  // tools.py:86-107, fdc.py:395-417
          cP = E.rz[g[a]];
          cM = E.rz[E.rz_n + g[a]];
          cB = E.rz[2 * E.rz_n + g[a]];
        }
        int rc = pa_row_case(G, a, g[a], N[a], G.treat);
        if (rc == 1) { cP = cB; cC = -cB; cM = (T)0; }
        if (rc == 2) { cP = (T)0; cC = -cB; cM = cB; }
        T xp, xm;
        pa_nbrs<T>(G, acc, a, i, j, k, xp, xm);
        T s = cP * xp;
        T m = cC * xc;
        s = s + m;
        m = cM * xm;
        s = s + m;
        ax = ax + s;
      }
      if (t.has_coeff) ax = ax * (t.coeff_f ? t.coeff_f[o] : t.coeff);
    } else if (t.kind == 1) {  // PA_OP_GRAD (1-D in the solver; sums axes otherwise never reached)
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        if (!G.act[a]) continue;
        T cP = E.grd.g[a], cC = (T)0, cM = E.grd.mg[a];
        int rc = pa_row_case(G, a, g[a], N[a], G.treat);
        if (rc == 1) { cP = E.grd.lo_p[a]; cC = E.grd.lo_c[a]; cM = (T)0; }
        if (rc == 2) { cP = (T)0; cC = E.grd.hi_c[a]; cM = E.grd.hi_m[a]; }
        if (G.bct[2 * a] == 4 && g[a] == 1) cM = (T)0;             // periodic lower (fdc.py:596-599)
        if (G.bct[2 * a + 1] == 4 && g[a] == N[a] - 2) cP = (T)0;  // periodic upper (fdc.py:600-602)
        T xp, xm;
        pa_nbrs<T>(G, acc, a, i, j, k, xp, xm);
        T s = cP * xp;
        T m = cC * xc;
        s = s + m;
        m = cM * xm;
        s = s + m;
        ax = ax + s;
      }
      if (t.has_coeff) ax = ax * (t.coeff_f ? t.coeff_f[o] : t.coeff);
    } else if (t.kind == 2) {  // PA_OP_DIV_CENTRAL (fdc.py:708-743), scalar phi: adv[0] on every axis
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        if (!G.act[a]) continue;
        T up = t.u, um = t.u;
        if (t.u_f) {
          // wrap-around neighbours of u on this rank (axis-0 ghosts of u are not exchanged:
          // the host restricts tensor-u central Div to P = 1)
          int64_t ii = i, jj = j, kk = k, i2 = i, j2 = j, k2 = k;
          if (a == 0) { ii = pa_wrap(i + 1, G.n0); i2 = pa_wrap(i - 1, G.n0); }
          if (a == 1) { jj = pa_wrap(j + 1, G.n1); j2 = pa_wrap(j - 1, G.n1); }
          if (a == 2) { kk = pa_wrap(k + 1, G.n2); k2 = pa_wrap(k - 1, G.n2); }
          up = t.u_f[ii * G.s0 + jj * G.s1 + kk];
          um = t.u_f[i2 * G.s0 + j2 * G.s1 + k2];
        }
        T ucen = t.u_f ? t.u_f[o] : t.u;
        T cP = up, cC = (T)0 * ucen, cM = -um;
        if (E.rz && a == PA_RZ_AXIS) cC = E.rz[4 * E.rz_n + g[a]] * ucen;  // tools.py:64-78
        if (G.bct[2 * a] == 4 && g[a] == 1) cM = (T)0;
        if (G.bct[2 * a + 1] == 4 && g[a] == N[a] - 2) cP = (T)0;
        cP = cP / E.grd.h2[a];
        cC = cC / E.grd.h2[a];
        cM = cM / E.grd.h2[a];
        T xp, xm;
        pa_nbrs<T>(G, acc, a, i, j, k, xp, xm);
        T s = cP * xp;
        T m = cC * xc;
        s = s + m;
        m = cM * xm;
        s = s + m;
        ax = ax + s;
      }
    } else if (t.kind == 3) {  // PA_OP_DIV_UPWIND_COMPAT (literal fdc.py:746-772)
      T ucen = t.u_f ? t.u_f[o] : t.u;
      T cP = (T)2 * (ucen < (T)0 ? ucen : (T)0);
      T cC0 = (T)0 * ((T)2 * ucen);
      T cM = (T)2 * (ucen > (T)0 ? ucen : (T)0);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        if (!G.act[a]) continue;
        T cC = cC0;
        if (E.rz && a == PA_RZ_AXIS) cC = E.rz[4 * E.rz_n + g[a]] * ((T)2 * ucen);
        T xp, xm;
        pa_nbrs<T>(G, acc, a, i, j, k, xp, xm);
        T s = cP * xp;
        T m = cC * xc;
        s = s + m;
        m = cM * xm;
        s = s + m;
        ax = ax + s;
      }
    } else {  // PA_OP_DIV_UPWIND (tests/test_fdm.py:239)
      T ucen = t.u_f ? t.u_f[o] : t.u;
      T upl = ucen > (T)0 ? ucen : (T)0;
      T umi = ucen < (T)0 ? ucen : (T)0;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        if (!G.act[a]) continue;
        T xp, xm;
        pa_nbrs<T>(G, acc, a, i, j, k, xp, xm);
        T bwd = xc - xm;
        T fwd = xp - xc;
        T s = upl * bwd;
        T m = umi * fwd;
        s = s + m;
        s = s * E.grd.ih[a];
        ax = ax + s;
      }
      if (E.rz) {  // + u phi / r, written like the central scheme's Ac row
        T cC = E.rz[4 * E.rz_n + g[PA_RZ_AXIS]] * ucen;
        cC = cC / E.grd.h2[PA_RZ_AXIS];
        T m = cC * xc;
        ax = ax + m;
      }
    }
    ax = ax * t.sign;
    res = res + ax;
  }
  return res;
}

// ---- reductions: wave64 shuffle, then across the block's waves through LDS -------------
template <int NS>
__device__ __forceinline__ void pa_block_reduce_store(double (&v)[NS], double* __restrict__ partials) {
  __shared__ double sm[NS][PA_BLOCK / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    double x = v[s];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if (lane == 0) sm[s][wave] = x;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      double x = sm[s][0];
      for (int w = 1; w < (int)(blockDim.x >> 6); ++w) x += sm[s][w];
      partials[(int64_t)blockIdx.x * NS + s] = x;
    }
  }
}
