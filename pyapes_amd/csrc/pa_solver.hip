// pa_solver.hip -- CG / Jacobi / BiCGSTAB drivers (linalg.py:33-279) with device-resident scalars, their
// generic kernels (any dimension / term list; the tiled kernels of pa_cg3d*.hip take over where they
// apply), the single-block reduction + scalar-step kernels, the stepwise CG entry points.
#include "pa_host.h"
#include "pa_scalar_steps.h"

#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>

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

// the same for the PITCH layout of the tiled CG phases (pa_cg3d_kernel.h, odd row lengths): A x arrives in a
// contiguous scratch, r and d leave with a row pitch of ps1 cells, pad cells zero.  Same loop, grid and partial sums.
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_cg_init_ax_pitch(DevGeom G, const T* __restrict__ rhs, const T* __restrict__ ax,
                                                                T* __restrict__ r, T* __restrict__ d, int64_t ps1,
                                                                double* __restrict__ partials) {
  double s[1] = {0.0};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t idx = t0; idx < G.ncell; idx += stride) {
    int64_t i, j, k;
    pa_decode(G, idx, i, j, k);
    T rv = (T)0;
    if (pa_in_S(G, i, j, k)) {
      rv = rhs[idx] - ax[idx];
      T p = rv * rv;
      s[0] += (double)p;
    }
    const int64_t o = (i * G.n1 + j) * ps1 + k;
    r[o] = rv;
    d[o] = rv;
  }
  const int64_t rows = G.n0 * G.n1, pw = ps1 - G.n2;
  for (int64_t q = t0; q < rows * pw; q += stride) {
    const int64_t o = (q / pw) * ps1 + G.n2 + q % pw;
    r[o] = (T)0;
    d[o] = (T)0;
  }
  pa_block_reduce_store<1>(s, partials);
}

// ... and with A x from the GENERIC term evaluation (axisymmetric meshes: no tiled A x), r / d (or r0 / r) pitched
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_cg_init_pitch(DevGeom G, DevEq<T> E, Vec<T> xv, const T* __restrict__ rhs,
                                                             T* __restrict__ r, T* __restrict__ d, int64_t ps1,
                                                             double* __restrict__ partials) {
  FieldAcc<T> acc{xv};
  double s[1] = {0.0};
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t idx = t0; idx < G.ncell; idx += stride) {
    int64_t i, j, k;
    pa_decode(G, idx, i, j, k);
    T rv = (T)0;
    if (pa_in_S(G, i, j, k)) {
      T ax = pa_apply_terms<T>(G, E, acc, i, j, k, xv.p[idx]);
      rv = rhs[idx] - ax;
      T p = rv * rv;
      s[0] += (double)p;
    }
    const int64_t o = (i * G.n1 + j) * ps1 + k;
    r[o] = rv;
    d[o] = rv;
  }
  const int64_t rows = G.n0 * G.n1, pw = ps1 - G.n2;
  for (int64_t q = t0; q < rows * pw; q += stride) {
    const int64_t o = (q / pw) * ps1 + G.n2 + q % pw;
    r[o] = (T)0;
    d[o] = (T)0;
  }
  pa_block_reduce_store<1>(s, partials);
}

// ---- CG phase A: d' = r + beta d ; partial sum d'.(A d')  (linalg.py:115-120, 141) ----
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_cg_a(DevGeom G, DevEq<T> E, const SolverScalars* __restrict__ sc,
                                                    Vec<T> rv, Vec<T> dv, T* __restrict__ dnew,
                                                    double* __restrict__ partials) {
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
}

// ---- CG phase B: x += alpha d ; r -= alpha A d ; partial sums r.r and |dx|^2 off-shell
//      (linalg.py:122-134)
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_cg_b(DevGeom G, DevEq<T> E, const SolverScalars* __restrict__ sc,
                                                    Vec<T> dv, T* __restrict__ x, const T* r, T* r_out,
                                                    T* __restrict__ send_lo, T* __restrict__ send_hi,
                                                    double* __restrict__ partials) {
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
      T p = rn * rn;
      s[0] += (double)p;
      if (!pa_on_shell(G, i, j, k)) {
        T df = xn - xo;
        T p2 = df * df;
        s[1] += (double)p2;
      }
    }
    // (every node: r itself, or r's new block when the placement search moves it -- 0 outside S, as r is everywhere)
    r_out[idx] = rn;
    if (send_lo && i == 0) send_lo[j * G.s1 + k] = rn;
    if (send_hi && i == G.n0 - 1) send_hi[j * G.s1 + k] = rn;
  }
  pa_block_reduce_store<2>(s, partials);
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
    pa_logic_jacobi<T>(sc, sums[2]);
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

// ---- slab, folded iteration: everything between phase A and phase B in ONE launch ------------------
// (i) alpha = r.r / d'.Ad' from the all-reduced d'.Ad' ROWS: every block sums them in the fixed order of
// k_cg_post_a / the phase-B prologue (same bits in every block, block 0 stores the state); (ii) the ghost
// planes of the new direction, d'_g = r_g + beta d_g (k_ghost_dir's recurrence); (iii) the planes the
// neighbours need from this iteration, computed AHEAD of phase B from the same operands in the same order
// (pa_apply_terms' Laplacian branch + k_cg_b's update, which the tiled phase B reproduces bit for bit): the
// new residual on the first / last owned plane and, on the end ranks of a periodic ring, the new x planes
// the other end's BC fill reads.  The packed exchange can therefore start before phase B and fly beside it.
// Folded iterations exist only where the tiled kernels run, i.e. for ONE Laplacian term on an xyz mesh, so
// the stencil is written out with all seven operands loaded up front (one memory round trip per node; the
// generic per-axis evaluation is a chain of three) -- a 512^2 plane pair: 29 us generic, ~10 us like this.
template <typename T>
struct MidArgs {
  const T* d;            // d' of this iteration
  const T* r;            // residual before phase B
  const T* x;            // iterate before phase B
  const T *r_lo, *r_hi;  // ghost planes of r (null: physical end)
  const T *d_lo, *d_hi;  // ghost planes of the PREVIOUS direction
  T *g_lo, *g_hi;        // out: ghost planes of d'
  T *send_lo, *send_hi;  // out: new residual on plane 0 / n0-1
  T *xp_lo1, *xp_hi0, *xp_hi1;  // out (periodic ring ends): new x on plane 1 / n0-1 / n0-2, or null
  LapCoef<T> lap;
  T coeff, sign;
  int has_coeff;
  const T* coeff_f;
};

// one axis of the Laplacian row at global index g: ((cP x+ + cC x) + cM x-), fdc.py:190-198 / pa_apply_terms
template <typename T>
__device__ __forceinline__ T pa_lap_axis(const DevGeom& G, const LapCoef<T>& L, int a, int64_t g, int64_t N, T xp, T xc,
                                         T xm) {
  T cP = L.inv[a], cC = L.m2inv[a], cM = L.inv[a];
  const T cB = L.c23[a];
  const int rc = pa_row_case(G, a, g, N, G.treat);
  if (rc == 1) { cP = cB; cC = -cB; cM = (T)0; }
  if (rc == 2) { cP = (T)0; cC = -cB; cM = cB; }
  T s = cP * xp;
  T m = cC * xc;
  s = s + m;
  m = cM * xm;
  s = s + m;
  return s;
}

template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_slab_mid(DevGeom G, SolverScalars* sc, const double* __restrict__ rows,
                                                        int nrows, double* __restrict__ sums, MidArgs<T> M) {
  __shared__ double sm[8];
  const int done_in = sc->done;
  const double rr_in = sc->rr;
  const T beta = (T)sc->beta;
  double v0 = 0.0;
  for (int b = threadIdx.x; b < nrows; b += PA_BLOCK) v0 += rows[b];
  if (done_in) return;
  for (int off = 32; off > 0; off >>= 1) v0 += __shfl_down(v0, off, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v0;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = 0.0;
    for (int w = 0; w < PA_BLOCK / 64; ++w) v += sm[w];
    const T dAd = (T)v;                      // linalg.py:118-120
    const T a = (T)rr_in / dAd;
    const double al = (isnan(a) || isinf(a)) ? 0.0 : (double)a;
    sm[4] = al;
    if (blockIdx.x == 0) {
      sc->dAd = (double)dAd;
      sc->alpha = al;
      sums[0] = v;
    }
  }
  __syncthreads();
  const T alpha = (T)sm[4];
  // gridDim.x = 5 sections x nb blocks (sections a rank does not have return at once)
  const int nb = (int)(gridDim.x / 5), sec = (int)(blockIdx.x / nb), bq = (int)(blockIdx.x - sec * nb);
  if (sec == 0 && !M.r_lo) return;
  if (sec == 1 && !M.r_hi) return;
  T* const xout = sec == 2 ? M.xp_lo1 : (sec == 3 ? M.xp_hi0 : M.xp_hi1);
  if (sec >= 2 && !xout) return;
  const int64_t ip = sec == 0 ? 0 : (sec == 1 ? G.n0 - 1 : (sec == 2 ? 1 : (sec == 3 ? G.n0 - 1 : G.n0 - 2)));
  const int64_t gi = ip + G.off0;
  const bool iS = gi >= G.slo[0] && gi <= G.shi[0];
  const T* const dpl = M.d + ip * G.s0;
  for (int64_t q = (int64_t)bq * blockDim.x + threadIdx.x; q < G.s0; q += (int64_t)nb * blockDim.x) {
    int64_t j, k;
    if (G.s0 <= 0x7fffffffLL) {
      const uint32_t qq = (uint32_t)q, s1 = (uint32_t)G.s1, jj = qq / s1;
      j = jj; k = qq - jj * s1;
    } else {
      j = q / G.s1; k = q - j * G.s1;
    }
    const bool inS = iS && j >= G.slo[1] && j <= G.shi[1] && k >= G.slo[2] && k <= G.shi[2];
    const T dc = dpl[q];
    if (sec >= 2) {   // new x on plane 1 / n0-1 / n0-2 (k_cg_b: x + alpha d' on S, x elsewhere)
      const T xo = M.x[ip * G.s0 + q];
      T ad = alpha * dc;
      T xn = xo + ad;
      xout[q] = inS ? xn : xo;
      continue;
    }
    // all operands of the node first: ghost pair, the plane inside, the four in-plane neighbours, r, Gamma
    const T rg = sec == 0 ? M.r_lo[q] : M.r_hi[q];
    const T dg = sec == 0 ? M.d_lo[q] : M.d_hi[q];
    const T din = sec == 0 ? M.d[G.s0 + q] : M.d[(G.n0 - 2) * G.s0 + q];
    const T dj1 = dpl[pa_wrap(j + 1, G.n1) * G.s1 + k], dj0 = dpl[pa_wrap(j - 1, G.n1) * G.s1 + k];
    const T dk1 = dpl[j * G.s1 + pa_wrap(k + 1, G.n2)], dk0 = dpl[j * G.s1 + pa_wrap(k - 1, G.n2)];
    const T rc = M.r[ip * G.s0 + q];
    T cf = M.coeff;
    if (M.has_coeff && M.coeff_f) cf = M.coeff_f[ip * G.s0 + q];
    T bg = beta * dg;
    const T ghost = rg + bg;                 // d'_g = r_g + beta d_g
    if (sec == 0) M.g_lo[q] = ghost; else M.g_hi[q] = ghost;
    // A d' at the node: axes 0 -> 1 -> 2 into zero, * Gamma, * sign, + 0 (pa_apply_terms, kind 0)
    T ax = (T)0;
    ax = ax + pa_lap_axis<T>(G, M.lap, 0, gi, G.g0, sec == 0 ? din : ghost, dc, sec == 0 ? ghost : din);
    ax = ax + pa_lap_axis<T>(G, M.lap, 1, j, G.n1, dj1, dc, dj0);
    ax = ax + pa_lap_axis<T>(G, M.lap, 2, k, G.n2, dk1, dc, dk0);
    if (M.has_coeff) ax = ax * cf;
    ax = ax * M.sign;
    T Ad = (T)0;
    Ad = Ad + ax;
    T aAd = alpha * Ad;
    T rn = rc - aAd;
    rn = inS ? rn : (T)0;
    if (sec == 0) M.send_lo[q] = rn; else M.send_hi[q] = rn;
  }
}

// x_old of the iteration that is about to update x (skipped, like the update, once the solve is over)
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_copy_guarded(const SolverScalars* __restrict__ sc, const T* __restrict__ a,
                                                            T* __restrict__ b, int64_t n) {
  if (sc->done) return;
  typedef T V __attribute__((ext_vector_type(16 / sizeof(T))));
  constexpr int VEC = 16 / sizeof(T);
  const bool vec = (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
  const int64_t nv = vec ? n / VEC : 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x)
    reinterpret_cast<V*>(b)[i] = reinterpret_cast<const V*>(a)[i];
  for (int64_t i = nv * VEC + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    b[i] = a[i];
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

#ifndef PA_BX_NTP
#define PA_BX_NTP 1   // k_bicg_x PITCHED: non-temporal accesses of the pitched (vector-aligned) arrays, as in the contiguous layout
#endif
#ifndef PA_BX_XU
#define PA_BX_XU 1    // k_bicg_x PITCHED: x in whole vectors at cell-aligned addresses (one 16-byte access for VEC 8- / 4-byte ones)
#endif
#ifndef PA_BX_XNT
#define PA_BX_XNT 1   // ... and those non-temporal
#endif
// early exit: x += alpha p ; otherwise x = x + alpha p + s omega ; r = s - omega t ; |r|^2
// VEC cells per lane and step (16-byte lane accesses where the four arrays allow them: 166 -> 1xx us at 256^3 fp64,
// DESIGN.md section 4; 1: any alignment / cell count)
// PITCHED (odd row lengths, bicg_run_t): p, s, t, r, v, p_next with a row pitch of ps1 cells (a multiple of the
// vector), x contiguous and touched cell by cell; pad cells are written as 0.
// SRV (round 4): s is not read but re-formed from r and v' -- s = r - alpha v', the combine of phase 6, operation for
// operation, so the same bits -- and r is updated IN PLACE: the tiled s / t phase then stores t alone (15 array passes per
// iteration for 16; s_in unused, v_in required).
template <typename T, int VEC, bool PITCHED = false, bool SRV = false>
__global__ void __launch_bounds__(PA_BLOCK) k_bicg_x(DevGeom G, const SolverScalars* __restrict__ sc,
                                                      T* __restrict__ x, const T* p,   // (p_next may be p: in place)
                                                      const T* __restrict__ s_in, const T* __restrict__ t_in,
                                                      T* __restrict__ r, double* __restrict__ partials,
                                                      const double* pre_part, int pre_n, SolverScalars* sc_w,
                                                      const T* v_in, T* p_next, int64_t ps1 = 0) {
  // p_next != null: also the NEXT direction p'' = r_new + beta (p' - omega v') (linalg.py:217) -- beta = rho_next / rho
  // alpha / omega is complete as soon as omega and rho_next = -omega (r0 . t) are (linalg.py:212, 246-247): the p / v
  // phase of the next iteration then reads ONE field with a halo instead of three and stores one instead of two
  // (v' = A p'' from the stored p'', phase 8 of k_cg3d); p'' goes unused when the stop test that follows ends the solve
  const T alpha = (T)sc->alpha;
  T omega;
  int early;
  T beta_n = (T)0;
  const double rho_cur = sc->rho;
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
      {   // the next beta, as the stage that closes the iteration forms it (k_bicg_post stage 3 / phase 5 prologue)
        T rn = -om;
        rn = rn * (T)t4[3];
        T bq = (T)(double)rn / (T)rho_cur;
        bq = bq * alpha;
        bq = bq / om;
        pre_sm[19] = (double)bq;
      }
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
    beta_n = (T)pre_sm[19];
  } else {
    if (sc->done) return;
    omega = (T)sc->omega;
    early = sc->finished_early;
    T bq = (T)sc->rho_next / (T)rho_cur;
    bq = bq * alpha;
    bq = bq / omega;
    beta_n = bq;
  }
  const bool pn = p_next != nullptr && !early;
  double s[1] = {0.0};
  typedef T V __attribute__((ext_vector_type(VEC)));
  const unsigned nvr = PITCHED ? (unsigned)(ps1 / VEC) : 1u;   // vectors per pitched row
  const int64_t nvec = PITCHED ? G.n0 * G.n1 * (int64_t)nvr : G.ncell / VEC;   // (VEC > 1 only for ncell % VEC == 0)
  // Traversal (round 4): every block owns ONE contiguous range of vectors and walks it backwards -- the s / t phase before
  // marched its chunks forwards, the v phase after will again, so what was touched last (still in the Infinity Cache) is
  // read first -- and the once-touched streams (x, r, t) move with non-temporal loads / stores.  The bare 5 : 3 mix at
  // 512^3 fp64 (profiles/tools/streammix2.hip): grid-stride 1.78-1.83 ms, contiguous ranges backwards + nt 1.64.
  constexpr bool NT = VEC > 1 && (!PITCHED || PA_BX_NTP);
  typedef T VU __attribute__((ext_vector_type(VEC), aligned(sizeof(T))));   // PITCHED: a vector of x at a cell-aligned address
  const int64_t per = ((nvec + gridDim.x - 1) / gridDim.x + PA_BLOCK - 1) / PA_BLOCK * PA_BLOCK;
  const int64_t b0 = (int64_t)blockIdx.x * per, b1 = b0 + per < nvec ? b0 + per : nvec;
  auto ldnt = [](const T* q, int64_t i) -> V {
    return NT ? __builtin_nontemporal_load(reinterpret_cast<const V*>(q) + i) : reinterpret_cast<const V*>(q)[i];
  };
  for (int64_t st = b1 > b0 ? (b1 - b0 + PA_BLOCK - 1) / PA_BLOCK - 1 : -1; st >= 0; --st) {
    const int64_t iv = b0 + st * PA_BLOCK + threadIdx.x;
    if (iv >= b1) continue;
    // (p and v' are read here for the last time in the iteration as well: non-temporal, which leaves the Infinity Cache to
    // the p'' this kernel writes for the v phase -- 256^3 fp64, eight interleaved pairs: 0.356-0.379 -> 0.351-0.353 ms / iteration)
    const V pv = ldnt(p, iv);
    V xv;
    T* xrow = nullptr;      // PITCHED: the cells of this vector in the caller's contiguous x
    int nval = VEC;         // ... and how many of them are real cells
    if (PITCHED) {
      const unsigned row = (unsigned)iv / nvr;
      const int64_t col = (int64_t)((unsigned)iv - row * nvr) * VEC;
      if (col >= G.n2) continue;   // a vector of pad cells: zero since the start of the solve, stays zero
      xrow = x + (int64_t)row * G.n2 + col;
      nval = (int)(G.n2 - col < VEC ? G.n2 - col : VEC);
      if (PA_BX_XU && VEC > 1 && nval == VEC) {
        xv = PA_BX_XNT ? __builtin_nontemporal_load(reinterpret_cast<const VU*>(xrow)) : *reinterpret_cast<const VU*>(xrow);
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) xv[v] = v < nval ? xrow[v] : (T)0;
      }
    } else {
      xv = ldnt(x, iv);
    }
    V xn, rn, sv, tv, vv, pq;
    if (!early) {
      if (SRV) {
        const V ro = ldnt(r, iv);
        vv = ldnt(v_in, iv);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          T av = alpha * vv[v];
          sv[v] = ro[v] - av;
        }
      } else {
        sv = reinterpret_cast<const V*>(s_in)[iv];
      }
      tv = ldnt(t_in, iv);
    }
    if (pn && !SRV) vv = ldnt(v_in, iv);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      T ap = alpha * pv[v];
      T xq = xv[v] + ap;
      if (!early) {
        T so = sv[v] * omega;
        xq = xq + so;
        T ot = omega * tv[v];
        T rq = sv[v] - ot;
        rn[v] = rq;
        T q = rq * rq;
        s[0] += (double)q;
        if (pn) {   // combine of k_cg3d phase 5
          T tq = omega * vv[v];
          tq = pv[v] - tq;
          tq = beta_n * tq;
          pq[v] = rq + tq;
          if (PITCHED && v >= nval) pq[v] = (T)0;
        }
      }
      xn[v] = xq;
    }
    if (pn) reinterpret_cast<V*>(p_next)[iv] = pq;
    if (!early) {
      if (NT) __builtin_nontemporal_store(rn, reinterpret_cast<V*>(r) + iv); else reinterpret_cast<V*>(r)[iv] = rn;
    }
    if (PITCHED) {
      if (PA_BX_XU && VEC > 1 && nval == VEC) {
        if (PA_BX_XNT) __builtin_nontemporal_store((VU)xn, reinterpret_cast<VU*>(xrow));
        else *reinterpret_cast<VU*>(xrow) = xn;
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v)
          if (v < nval) xrow[v] = xn[v];
      }
    } else if (NT) {
      __builtin_nontemporal_store(xn, reinterpret_cast<V*>(x) + iv);
    } else {
      reinterpret_cast<V*>(x)[iv] = xn;
    }
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

// The cells of r / d the tiled CG phases never write -- the last boundary row / column of a non-periodic axis
// (launch_cg3d does not tile them) and the pad cells of pitched rows -- must read 0 (pa_cg_begin zeroes them once).  A
// block the placement search brings in fresh from hipMalloc (pa_place.hip) needs exactly these zeroed, not a memset of
// the whole array: O(n^2) cells instead of a write pass (0.16 ms at 512^3, and nothing at all on a fully periodic mesh).
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_zero_skipped(DevGeom G, T* __restrict__ a, int64_t ps1, int zrow, int zcol) {
  const int64_t s1 = ps1 ? ps1 : G.s1, rows = G.n0 * G.n1, pw = ps1 ? ps1 - G.n2 : 0;
  const int64_t n_pad = rows * pw, n_row = zrow ? G.n0 * G.n2 : 0, n_col = zcol ? rows : 0;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_pad + n_row + n_col; q += (int64_t)gridDim.x * blockDim.x) {
    int64_t o;
    if (q < n_pad) {
      o = (q / pw) * s1 + G.n2 + q % pw;
    } else if (q < n_pad + n_row) {
      const int64_t t = q - n_pad, i = t / G.n2, k = t - i * G.n2;
      o = (i * G.n1 + G.n1 - 1) * s1 + k;
    } else {
      o = (q - n_pad - n_row) * s1 + G.n2 - 1;
    }
    a[o] = (T)0;
  }
}

// pa_place.hip: make `block` (the place of r or of a direction buffer in a block that has not carried one in this solve)
// fit for the phases; returns 1 if a kernel was enqueued
int pa_place_prepare_block(pa_ctx* c, void* block) {
  const DevGeom& G = c->G;
  const int zrow = G.act[1] && G.bct[3] != PA_BC_PERIODIC && G.n1 > 2, zcol = G.bct[5] != PA_BC_PERIODIC && G.n2 > 2;
  const int64_t ps1 = c->cg_pitch ? c->cg_ps1 : 0;
  const int64_t work = (ps1 ? G.n0 * G.n1 * (ps1 - G.n2) : 0) + (zrow ? G.n0 * G.n2 : 0) + (zcol ? G.n0 * G.n1 : 0);
  if (work <= 0) return 0;
  if (c->dtype == PA_F64)
    hipLaunchKernelGGL(k_zero_skipped<double>, dim3(pa_grid_blocks(work)), dim3(PA_BLOCK), 0, c->stream, G, (double*)block, ps1, zrow, zcol);
  else
    hipLaunchKernelGGL(k_zero_skipped<float>, dim3(pa_grid_blocks(work)), dim3(PA_BLOCK), 0, c->stream, G, (float*)block, ps1, zrow, zcol);
  return 1;
}

// ---- host side ----------------------------------------------------------------------------------
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

static int read_scalars(pa_ctx* c) {
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

// after a resident launch (pa_resident.hip): the solve has run to its end inside the one kernel -- returns 1 --
// or one of its bounded grid-wide waits timed out and the kernel left x, r and the scalars untouched -- returns 0:
// the caller goes on into its launch-per-phase loop as if the launch had not happened
static int resident_finish(pa_ctx* c, pa_report* out, int* rc_out) {
  int* h_fail = (int*)&c->h_poll[0]->rr;   // pinned scratch (the poll slots are idle here)
  *h_fail = 0;
  hipError_t e = hipMemcpyAsync(h_fail, (const char*)c->scr[SCR_RES] + 64, sizeof(int), hipMemcpyDeviceToHost, c->stream);
  if (e != hipSuccess) { *rc_out = pa_hip_fail(c, e, "resident fail flag"); return 1; }
  int rc = read_scalars(c);   // synchronises the stream
  if (rc) { c->solver_live = 0; *rc_out = rc; return 1; }
  if (*h_fail) {
    c->resident_used = 0;
    return 0;
  }
  c->solver_live = 0;
  e = hipEventRecord(c->ev1, c->stream);
  if (e == hipSuccess) e = hipEventSynchronize(c->ev1);
  if (e != hipSuccess) { *rc_out = pa_hip_fail(c, e, "resident finish"); return 1; }
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
  fill_report(c, out, ms);
  *rc_out = c->h_sc->err ? PA_E_NONFINITE : PA_OK;
  return 1;
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

// Row pitch (cells) of the ctx-owned solver arrays when the PITCH layout applies to a solve on x, else 0.
// Row lengths that are not a multiple of the 16-byte vector -- the normal case of a node-based mesh (11, 101,
// 2^k + 1 nodes: _mesh.py:67-93) -- on one GPU: the arrays the ctx owns get a row pitch rounded up (PITCH layout of
// k_cg3d), so the solver phases keep their 16-byte lane accesses on everything but the caller's x.  Needs a
// non-periodic contiguous axis (a pad cell must never be a neighbour that is used) and a plain Laplacian; everything
// else stays on the one-cell-per-lane (NARROW) kernels.
template <typename T>
static int64_t solver_pitch(const pa_ctx* c, const T* x) {
  const DevGeom& G = c->G;
  constexpr int VECW = 16 / (int)sizeof(T);
  const bool shape = (c->ndim == 3 && G.n0 >= 3 && G.n1 >= 3) || (c->ndim == 2 && G.n1 >= 3);
  // (axisymmetric meshes: their one tiled kernel is the 2-D marching k_cg2d<..., RZ>, pa_cg2d_kernel.h)
  const bool coord_ok = c->coord == PA_COORD_XYZ ||
                        (c->coord == PA_COORD_RZ && c->ndim == 2 && c->rz_tab && G.n1 >= 8 && c->cg2d_mincells >= 0 &&
                         G.n1 * G.n2 >= std::min<int64_t>(c->cg2d_mincells, 150000));
  if (!(c->pitch && c->fastpath && !c->slab && coord_ok && shape && G.n2 % VECW != 0 && G.n2 >= 2 * VECW &&
        c->nterms == 1 && c->terms[0].kind == PA_OP_LAPLACIAN && !c->terms[0].coeff_field &&
        G.bct[4] != PA_BC_PERIODIC && G.bct[5] != PA_BC_PERIODIC && ((uintptr_t)x & (sizeof(T) - 1)) == 0 &&
        ((G.n1 + 3) / 4) * ((G.n2 + 64 * VECW - 1) / (64 * VECW)) <= PA_MAX_PARTIALS))
    return 0;
  // pitch: a multiple of 128 bytes, not merely of the vector -- measured at 257^3 fp64 with 16-byte granularity
  // (258 cells): phase A 108 us against 79 at 256^3 on the same tiling, although every access was a vector:
  // rows that start inside a cache line make every tile edge a line shared by two workgroups, and their
  // streaming stores partial-line writes
  const int64_t padw = 128 / (int64_t)sizeof(T);
  return (G.n2 + padw - 1) / padw * padw;
}

template <typename T>
static int cg_begin_t(pa_ctx* c, T* x, const T* rhs, double tol, int64_t max_it) {
  const DevGeom& G = c->G;
  const size_t fb = (size_t)G.ncell * sizeof(T);
  const int nblk = pa_grid_blocks(G.ncell);
  int rc;
  // odd row lengths: r and the two direction buffers in the PITCH layout (solver_pitch)
  c->cg_ps1 = solver_pitch<T>(c, x);
  c->cg_pitch = c->cg_ps1 > 0 ? 1 : 0;
  const size_t fbp = c->cg_pitch ? (size_t)G.n0 * G.n1 * c->cg_ps1 * sizeof(T) : fb;
  if ((rc = pa_scratch(c, &c->scr[SCR_R], &c->cap[SCR_R], fbp))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_D0], &c->cap[SCR_D0], fbp))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_D1], &c->cap[SCR_D1], fbp))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_PART], &c->cap[SCR_PART], (size_t)PA_MAX_PARTIALS * 4 * sizeof(double)))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_PART2], &c->cap[SCR_PART2], (size_t)3 * PA_MAX_GRID * sizeof(double)))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_SHELL], &c->cap[SCR_SHELL], 2 * (size_t)pa_shell_elems(c) * sizeof(T)))) return rc;
  if ((rc = init_scalars(c, tol, max_it))) return rc;
  // large solves: the online search for the allocations r / d / d' should live in (pa_place.hip) rides on the iterations
  if ((rc = pa_place_begin(c, x, fb))) return rc;   // (before anything is written into r / d)
  // the tiled phase kernels do not visit the last boundary row / column of non-periodic axes: the
  // direction there is 0 by definition and has to be 0 in the buffer the first phase A writes into
  if (!c->cg_pitch) PA_HIP(c, hipMemsetAsync(c->scr[SCR_D1], 0, fb, c->stream));   // (pitched: it first carries A x, below)
  c->cg_x = x;
  c->cur = 0;
  c->fold_a_n = c->fold_b_n = c->fold_b_nsh = 0;  // nothing of an earlier (possibly failed) solve is pending
  c->fold_b_shell = nullptr;
  c->slab_fold = c->slab_fold_live = 0;           // row counts are agreed per solve (pa_cg_fold_plan / _set)
  c->bc_static = pa_bc_is_static(c);
  c->bc_fused = pa_bc_fusable(c);
  c->bc_pair = (!c->bc_fused && pa_bc_pairable(c)) ? 1 : 0;
  c->shell_cur = 0;
  c->solver_live = 1;
  DevEq<T> E;
  pa_build_eq<T>(c, c->nterms, c->terms, E);
  // linalg.py:97.  On a slab the driver fills the BCs itself (pa_apply_bc) BEFORE it exchanges
  // the ghost planes of x, so the fill must not run again here.
  bool shell_ready = false;
  if (!c->slab) {
    if (c->bc_fused) {  // fill + remember the filled shell as x_old in one go
      if ((rc = pa_bc_shell_fused<T>(c, x, nullptr, 0, false, nullptr, false))) return rc;
      shell_ready = true;
    } else if (c->bc_pair) {
      if ((rc = pa_bc_pair_apply<T>(c, x, nullptr, 2, false, nullptr))) return rc;
      shell_ready = true;
    } else if ((rc = pa_bc_apply_faces<T>(c, x))) {
      return rc;
    }
  } else if (c->bc_pair) {  // slab: the driver has filled the BCs already; only record the shell
    if ((rc = pa_bc_pair_apply<T>(c, x, nullptr, 3, false, nullptr))) return rc;
    shell_ready = true;
  }
  T* r = (T*)c->scr[SCR_R];
  T* d = (T*)c->scr[SCR_D0];
  double* part = (double*)c->scr[SCR_PART];
  Vec<T> xv = pa_vec_self<T>(c, x);
  if (c->slab) { xv.glo = (const T*)c->x_glo; xv.ghi = (const T*)c->x_ghi; }
  if (c->cg_pitch) {
    // A x (tiled kernel, contiguous) into the buffer that becomes the zeroed second direction buffer afterwards
    T* ax = (T*)c->scr[SCR_D1];
    const int fr = pa_tile3d_aop<T>(c, E, xv, ax, 1);
    if (fr < 0) return fr;
    if (fr > 0) {
      hipLaunchKernelGGL(k_cg_init_ax_pitch<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, c->G, rhs, (const T*)ax, r, d,
                         c->cg_ps1, part);
    } else if (c->coord == PA_COORD_RZ) {   // no tiled A x with r rows: the generic term evaluation, once per solve
      hipLaunchKernelGGL(k_cg_init_pitch<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, c->G, E, xv, rhs, r, d, c->cg_ps1, part);
    } else {
      c->cg_pitch = 0;   // the tiled A x declined: contiguous layout, as before (the buffers are merely larger)
      c->cg_ps1 = 0;
    }
    PA_HIP(c, hipMemsetAsync(c->scr[SCR_D1], 0, fbp, c->stream));
  }
  if (!c->cg_pitch && (rc = cg_residual_init<T>(c, E, xv, rhs, r, d, (T*)c->r_send_lo, (T*)c->r_send_hi, part))) return rc;
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
    pa_shell_launch<T>(c, (const T*)x, (T*)c->scr[SCR_SHELL] + (c->shell_cur ? pa_shell_elems(c) : 0), (double*)c->scr[SCR_PART2], 0);
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

// profiling build of the launch path: HIP events on the ctx stream bracket exactly one
// dominant kernel; the host waits for each, so use it in a dedicated measurement loop only
void pa_profile_stop(pa_ctx* c, int which) {
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
  if (c->cg_pitch) v.glo = p + (c->G.n0 - 1) * c->G.n1 * c->cg_ps1;   // the wrap-around plane of a pitched array
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
                       c->fold_b_n, c->fold_b_shell ? c->fold_b_shell : (const double*)c->scr[SCR_PART2],
                       c->fold_b_nsh, pa_sums(c), 2);
    c->fold_b_n = c->fold_b_nsh = 0;
    c->fold_b_shell = nullptr;
  }
}

template <typename T>
int pa_cg_phase_a_t(pa_ctx* c, int stage_post) {
  PaRange range_("pyapes CG phase A: d' = r + beta d, sum d'.(A d')");
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
  const bool foldable = c->fold && c->in_iterate && stage_post == 2 && !c->slab && !c->profile;
  if (foldable) part += 2 * (size_t)PA_MAX_PARTIALS;
  const bool live = c->slab_fold_live != 0;   // folded slab iteration: rows go out through the all-reduce buffer
  if (live) part = c->rows_send;
  if (c->pending_init_logic) {  // slab: sum r.r has been all-reduced by the driver
    hipLaunchKernelGGL(k_cg_post_init<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, (const double*)nullptr, 0,
                       pa_sums(c), 1);
    c->pending_init_logic = 0;
  }
  Vec<T> rv = cg_vec<T>(c, r, 0), dv = cg_vec<T>(c, dold, 1);
  if (c->slab && !live && (c->r_recv_lo || c->r_recv_hi)) {
    hipLaunchKernelGGL(k_ghost_dir<T>, dim3(pa_grid_blocks(G.s0)), dim3(PA_BLOCK), 0, c->stream, c->sc, G.s0,
                       (const T*)c->r_recv_lo, (const T*)c->r_recv_hi, (const T*)c->d_glo[c->cur],
                       (const T*)c->d_ghi[c->cur], (T*)c->d_glo[c->cur ^ 1], (T*)c->d_ghi[c->cur ^ 1]);
  }
  if (c->profile) (void)hipEventRecord(c->pev[0], c->stream);
  int rc = pa_cg3d_phase_a<T>(c, E, rv, dv, dnew, part);
  if (rc < 0) return rc;
  int used_blocks = rc;
  if (rc == 0 && live) { pa_set_err(c, "folded slab iteration: the tiled phase A declined after the plan"); return PA_E_STATE; }
  if (rc == 0 && c->cg_pitch) { pa_set_err(c, "pitched CG: the tiled phase A declined"); return PA_E_STATE; }
  if (rc == 0) {
    cg_flush_fold<T>(c);  // the tiled kernel declined: the previous iteration is closed by its own kernel
    hipLaunchKernelGGL(k_cg_a<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, G, E, c->sc, rv, dv, dnew, part);
    used_blocks = nblk;
  }
  if (c->profile) pa_profile_stop(c, 0);
  c->cur ^= 1;
  if (live) {
    // alpha comes from the mid kernel, after the all-reduce of the rows
  } else if (foldable && used_blocks <= PA_MAX_GRID)
    c->fold_a_n = used_blocks;  // phase B's prologue (or cg_flush_fold) computes alpha
  else
    hipLaunchKernelGGL(k_cg_post_a<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, part, used_blocks, pa_sums(c),
                       stage_post);
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

template <typename T>
int pa_cg_phase_b_t(pa_ctx* c, int stage_post) {
  PaRange range_("pyapes CG phase B: x += alpha d', r -= alpha A d', BC fill, sums");
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
  const bool live = c->slab_fold_live != 0;
  if (live) part = c->rows_send + c->fold_rows[0] + c->fold_rows[2];
  if (c->profile) (void)hipEventRecord(c->pev[2], c->stream);
  int rc = pa_cg3d_phase_b<T>(c, E, dv, x, r, part);
  if (rc < 0) return rc;
  int used_blocks = rc;
  if (rc == 0 && live) { pa_set_err(c, "folded slab iteration: the tiled phase B declined after the plan"); return PA_E_STATE; }
  if (rc == 0 && c->cg_pitch) { pa_set_err(c, "pitched CG: the tiled phase B declined"); return PA_E_STATE; }
  if (rc == 0) {
    cg_flush_fold<T>(c);  // alpha by its own kernel
    hipLaunchKernelGGL(k_cg_b<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, G, E, c->sc, dv, x, (const T*)r,
                       c->cg_r_out ? (T*)c->cg_r_out : r, (T*)c->r_send_lo, (T*)c->r_send_hi, part);
    used_blocks = nblk;
  }
  if (c->cg_r_out) pa_place_r_written(c);   // the placement search moved r with this launch: SCR_R is the new block now
  if (c->profile) pa_profile_stop(c, 1);
  c->b_blocks = used_blocks;
  if (c->slab) {  // BC fill + shell + reduction happen in pa_cg_bc, after the driver's plane exchange
    if (!live && (c->x_pack_lo1 || c->x_pack_hi0 || c->x_pack_hi1)) {
      const T* xr = (const T*)x;
      hipLaunchKernelGGL(k_pack_planes<T>, dim3(pa_grid_blocks(G.s0)), dim3(PA_BLOCK), 0, c->stream, c->sc, G.s0,
                         xr + 1 * G.s0, (T*)c->x_pack_lo1, xr + (G.n0 - 1) * G.s0, (T*)c->x_pack_hi0,
                         xr + (G.n0 - 2) * G.s0, (T*)c->x_pack_hi1);
    }
    PA_HIP(c, hipGetLastError());
    return PA_OK;
  }
  int nsh = 0;
  if (!c->bc_static) {
    if (c->bc_fused) {
      if ((rc = pa_bc_shell_fused<T>(c, x, part2, 1, true, &nsh, false))) return rc;
    } else if (c->bc_pair) {
      if ((rc = pa_bc_pair_apply<T>(c, x, part2, 1, true, &nsh))) return rc;
    } else {
      if ((rc = pa_bc_apply_faces<T>(c, x, true))) return rc;
      nsh = pa_shell_blocks(c);
      pa_shell_launch<T>(c, (const T*)x, (T*)c->scr[SCR_SHELL], part2, 1);
    }
  }
  const bool foldable = c->fold && c->in_iterate && stage_post == 2 && !c->slab && !c->profile &&
                        used_blocks <= PA_MAX_GRID && nsh <= 3 * PA_MAX_GRID;
  if (foldable) {
    c->fold_b_n = used_blocks;  // the next phase A's prologue (or cg_flush_fold) closes this iteration
    c->fold_b_nsh = nsh;
    c->fold_b_part = part;
  } else {
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
  const bool live = c->slab_fold_live != 0;
  if (live) part2 = c->rows_send + c->fold_rows[0];
  int nsh = 0, rc;
  if (!c->bc_static) {
    if (c->bc_fused) {
      if ((rc = pa_bc_shell_fused<T>(c, x, part2, 1, true, &nsh, false))) return rc;
    } else if (c->bc_pair) {
      if ((rc = pa_bc_pair_apply<T>(c, x, part2, 1, true, &nsh))) return rc;
    } else {
      if ((rc = pa_bc_apply_faces<T>(c, x, true))) return rc;
      nsh = pa_shell_blocks(c);
      pa_shell_launch<T>(c, (const T*)x, (T*)c->scr[SCR_SHELL], part2, 1);
    }
  }
  if (live) {
    // the all-reduced rows are summed by the next phase A's prologue (or pa_cg_slab_flush)
    if (nsh > c->fold_rows[2]) { pa_set_err(c, "folded slab iteration: %d shell rows, %d planned", nsh, c->fold_rows[2]); return PA_E_STATE; }
    c->fold_b_part = c->rows_recv + c->fold_rows[0] + c->fold_rows[2];
    c->fold_b_n = c->fold_rows[1];
    c->fold_b_shell = c->rows_recv + c->fold_rows[0];
    c->fold_b_nsh = c->fold_rows[2];
  } else {
    hipLaunchKernelGGL(k_cg_post_b<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, part, c->b_blocks, part2, nsh,
                       pa_sums(c), 0);
  }
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

// folded slab iteration, between the all-reduce of the d.Ad rows and phase B (k_slab_mid)
template <typename T>
static int cg_slab_mid_t(pa_ctx* c) {
  const DevGeom& G = c->G;
  DevEq<T> E;
  pa_build_eq<T>(c, c->nterms, c->terms, E);
  if (E.nterms != 1 || E.t[0].kind != PA_OP_LAPLACIAN || c->coord != PA_COORD_XYZ || !G.act[0]) {
    pa_set_err(c, "folded slab iteration: one Laplacian term on a 3-D xyz mesh expected (the tiled kernels' equation)");
    return PA_E_STATE;
  }
  MidArgs<T> M;
  memset(&M, 0, sizeof(M));
  M.lap = E.lap;
  M.coeff = E.t[0].coeff; M.sign = E.t[0].sign; M.has_coeff = E.t[0].has_coeff; M.coeff_f = E.t[0].coeff_f;
  M.d = (const T*)c->scr[c->cur ? SCR_D1 : SCR_D0];
  M.r = (const T*)c->scr[SCR_R];
  M.x = (const T*)c->cg_x;
  M.r_lo = (const T*)c->r_recv_lo; M.r_hi = (const T*)c->r_recv_hi;
  M.d_lo = (const T*)c->d_glo[c->cur ^ 1]; M.d_hi = (const T*)c->d_ghi[c->cur ^ 1];
  M.g_lo = (T*)c->d_glo[c->cur]; M.g_hi = (T*)c->d_ghi[c->cur];
  M.send_lo = (T*)c->r_send_lo; M.send_hi = (T*)c->r_send_hi;
  if ((M.r_lo && !M.send_lo) || (M.r_hi && !M.send_hi)) { pa_set_err(c, "slab: a neighbour without a send plane"); return PA_E_STATE; }
  M.xp_lo1 = (T*)c->x_pack_lo1; M.xp_hi0 = (T*)c->x_pack_hi0; M.xp_hi1 = (T*)c->x_pack_hi1;
  const bool planes = M.r_lo || M.r_hi || M.xp_lo1 || M.xp_hi0 || M.xp_hi1;
  // <= 256 blocks per section: with five sections the whole grid is resident at once (a second round of
  // blocks would pay the prologue's round trip again)
  const int nbm = planes ? std::min(256, pa_grid_blocks(G.s0)) : 1;
  hipLaunchKernelGGL(k_slab_mid<T>, dim3(5 * nbm), dim3(PA_BLOCK), 0, c->stream, G, c->sc,
                     (const double*)c->rows_recv, c->fold_rows[0], pa_sums(c), M);
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

int pa_cg_slab_mid(pa_ctx* c) {
  return c->dtype == PA_F64 ? cg_slab_mid_t<double>(c) : cg_slab_mid_t<float>(c);
}

int pa_cg_slab_flush(pa_ctx* c) {
  if (c->dtype == PA_F64) cg_flush_fold<double>(c); else cg_flush_fold<float>(c);
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

template <typename T>
static int cg_run_t(pa_ctx* c, T* x, const T* rhs, double tol, int64_t max_it, pa_report* out) {
  // small meshes: the whole solve -- set-up included -- in one cooperative launch (pa_resident.hip)
  PA_HIP(c, hipEventRecord(c->ev0, c->stream));
  c->resident_used = pa_resident_launch<T>(c, 0, x, rhs, tol, max_it, 1.0);
  if (c->resident_used < 0) return c->resident_used;
  if (c->resident_used > 0) {
    int rrc = PA_OK;
    if (resident_finish(c, out, &rrc)) return rrc;
  }
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
      if ((rc = pa_place_tick(c))) break;
      if ((rc = pa_cg_phase_a_t<T>(c, 2))) break;
      if (c->x_old_out)   // after phase A: its prologue has decided whether this iteration still runs
        hipLaunchKernelGGL(k_copy_guarded<T>, dim3(pa_grid_blocks(c->G.ncell)), dim3(PA_BLOCK), 0, c->stream, c->sc,
                           (const T*)x, (T*)c->x_old_out, c->G.ncell);
      rc = pa_cg_phase_b_t<T>(c, 2);
      ++enq;
    }
    if (rc) break;
    if ((rc = pa_place_batch_end(c))) break;
    cg_flush_fold<T>(c);
    rc = poll_submit(c, P, &done);
    batch = std::min<int64_t>(poll, std::max<int64_t>(1, enq));
  }
  if (!rc) rc = read_scalars(c);
  c->in_iterate = 0;
  if (rc) { c->fold_a_n = c->fold_b_n = c->fold_b_nsh = 0; pa_place_end(c, 0); return rc; }
  PA_HIP(c, hipEventRecord(c->ev1, c->stream));
  PA_HIP(c, hipEventSynchronize(c->ev1));
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
  fill_report(c, out, ms);
  c->solver_live = 0;
  pa_place_end(c, 1);
  return c->h_sc->err ? PA_E_NONFINITE : PA_OK;
}

template <typename T>
static int jacobi_run_t(pa_ctx* c, T* x, const T* rhs, double tol, int64_t max_it, double omega, pa_report* out) {
  const DevGeom& G = c->G;
  for (int q = 0; q < c->nterms; ++q)
    if (c->terms[q].kind != PA_OP_LAPLACIAN) { pa_set_err(c, "pa_jacobi: laplacian terms only"); return PA_E_ARG; }
  if (c->slab) { pa_set_err(c, "pa_jacobi is single-GPU only"); return PA_E_ARG; }
  PA_HIP(c, hipEventRecord(c->ev0, c->stream));
  c->resident_used = pa_resident_launch<T>(c, 1, x, rhs, tol, max_it, omega);   // small meshes: pa_resident.hip
  if (c->resident_used < 0) return c->resident_used;
  if (c->resident_used > 0) {
    int rrc = PA_OK;
    if (resident_finish(c, out, &rrc)) return rrc;
  }
  const size_t fb = (size_t)G.ncell * sizeof(T);
  const int nblk = pa_grid_blocks(G.ncell);
  int rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_D0], &c->cap[SCR_D0], fb))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_PART], &c->cap[SCR_PART], (size_t)PA_MAX_PARTIALS * 4 * sizeof(double)))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_PART2], &c->cap[SCR_PART2], (size_t)3 * PA_MAX_GRID * sizeof(double)))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_SHELL], &c->cap[SCR_SHELL], 2 * (size_t)pa_shell_elems(c) * sizeof(T)))) return rc;
  if ((rc = init_scalars(c, tol, max_it))) return rc;
  DevEq<T> E;
  pa_build_eq<T>(c, c->nterms, c->terms, E);
  const bool stat = pa_bc_is_static(c);
  double* part = (double*)c->scr[SCR_PART];
  double* part2 = (double*)c->scr[SCR_PART2];
  // BC fill by the cheapest launch sequence, as in CG: closed form / one launch per axis / one per face
  c->bc_fused = pa_bc_fusable(c);
  c->bc_pair = (!c->bc_fused && pa_bc_pairable(c)) ? 1 : 0;
  c->shell_cur = 0;
  c->fold_a_n = c->fold_b_n = c->fold_b_nsh = 0;
  if (c->bc_fused) {
    if ((rc = pa_bc_shell_fused<T>(c, x, nullptr, 0, false, nullptr, false))) return rc;
  } else if (c->bc_pair) {
    if ((rc = pa_bc_pair_apply<T>(c, x, nullptr, 2, false, nullptr))) return rc;
  } else {
    if ((rc = pa_bc_apply_faces<T>(c, x))) return rc;
    pa_shell_launch<T>(c, (const T*)x, (T*)c->scr[SCR_SHELL], part2, 0);
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
          if ((rc = pa_bc_shell_fused<T>(c, buf[cur ^ 1], part2, 1, true, &nsh, false))) return rc;
        } else if (c->bc_pair) {
          if ((rc = pa_bc_pair_apply<T>(c, buf[cur ^ 1], part2, 1, true, &nsh))) return rc;
        } else {
          if ((rc = pa_bc_apply_faces<T>(c, buf[cur ^ 1], true))) return rc;
          nsh = pa_shell_blocks(c);
          pa_shell_launch<T>(c, (const T*)buf[cur ^ 1], (T*)c->scr[SCR_SHELL], part2, 1);
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
  // the final iterate lives in buf[itr & 1], the one before it (Field.VARo) in the other buffer
  if (c->x_old_out && c->h_sc->itr >= 1)
    hipLaunchKernelGGL(k_copy<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, (const T*)buf[(c->h_sc->itr & 1) ^ 1],
                       (T*)c->x_old_out, G.ncell);
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
  PA_HIP(c, hipEventRecord(c->ev0, c->stream));
  c->resident_used = pa_resident_launch<T>(c, 2, x, rhs, tol, max_it, 1.0);   // small meshes: pa_resident.hip
  if (c->resident_used < 0) return c->resident_used;
  if (c->resident_used > 0) {
    int rrc = PA_OK;
    if (resident_finish(c, out, &rrc)) return rrc;
  }
  const int nblk = pa_grid_blocks(G.ncell);
  int rc;
  // odd row lengths (round 3): ALL eight solver arrays are the ctx's, so all of them take the PITCH layout
  // (solver_pitch) and the tiled phases keep their 16-byte lanes; only the x / r update touches the caller's x.
  // (an index of vectors must fit 31 bits in k_bicg_x)
  // Measured (us / iteration, one-cell-per-lane -> pitched, same box): 257^3 fp64 498-503 -> 470 (256^3: 430), 2-D 4097^2
  // 574 -> 503, 1025^2 59 -> 50, 129^3 76 -> 78; 257^3 fp32 274 -> 287 -- so: fp64, or a 2-D mesh.
  c->cg_ps1 = (sizeof(T) == 8 || c->ndim == 2) ? solver_pitch<T>(c, x) : 0;
  if (c->cg_ps1 > 0 && G.n0 * G.n1 * (c->cg_ps1 / (16 / (int64_t)sizeof(T))) >= ((int64_t)1 << 31)) c->cg_ps1 = 0;
  c->cg_pitch = c->cg_ps1 > 0 ? 1 : 0;
  struct PitchOff { pa_ctx* c; ~PitchOff() { c->cg_pitch = 0; c->cg_ps1 = 0; } } pitch_off{c};   // (a CG solve sets its own)
  const size_t fb = c->cg_pitch ? (size_t)G.n0 * G.n1 * c->cg_ps1 * sizeof(T) : (size_t)G.ncell * sizeof(T);
  const int ids[] = {SCR_R, SCR_D0, SCR_D1, SCR_R0, SCR_V0, SCR_V1, SCR_S, SCR_TT};
  for (int id : ids)
    if ((rc = pa_scratch(c, &c->scr[id], &c->cap[id], fb))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_PART], &c->cap[SCR_PART], (size_t)PA_MAX_PARTIALS * 6 * sizeof(double)))) return rc;
  if ((rc = init_scalars(c, tol, max_it))) return rc;
  DevEq<T> E;
  pa_build_eq<T>(c, c->nterms, c->terms, E);
  if ((rc = pa_bc_apply_faces<T>(c, x))) return rc;
  T* r = (T*)c->scr[SCR_R];
  T* r0 = (T*)c->scr[SCR_R0];
  T* p[2] = {(T*)c->scr[SCR_D0], (T*)c->scr[SCR_D1]};
  T* v[2] = {(T*)c->scr[SCR_V0], (T*)c->scr[SCR_V1]};
  T* s = (T*)c->scr[SCR_S];
  T* t = (T*)c->scr[SCR_TT];
  double* part = (double*)c->scr[SCR_PART];
  Vec<T> xv = pa_vec_self<T>(c, x);
  // a field of the solver as the tiled phases see it (wrap-around planes of a pitched array: its own)
  auto vec_of = [&](const T* q) -> Vec<T> {
    Vec<T> w = pa_vec_self<T>(c, q);
    if (c->cg_pitch) { w.glo = q + (G.n0 - 1) * G.n1 * c->cg_ps1; w.ghi = q; }
    return w;
  };
  if (c->cg_pitch) {
    // A x (tiled kernel, contiguous) into t, then r0 = r = b - A x scattered into the pitched rows with the loop and
    // partial sums of the contiguous form (k_cg_init_ax_pitch); pad cells of every array zero for the whole solve
    const int fr = pa_tile3d_aop<T>(c, E, xv, t, 1);
    if (fr < 0) return fr;
    if (fr > 0) {
      hipLaunchKernelGGL(k_cg_init_ax_pitch<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, c->G, rhs, (const T*)t, r0, r,
                         c->cg_ps1, part);
    } else if (c->coord == PA_COORD_RZ) {
      hipLaunchKernelGGL(k_cg_init_pitch<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, c->G, E, xv, rhs, r0, r, c->cg_ps1, part);
    } else {   // the tiled A x declined: contiguous layout (the buffers are merely larger)
      c->cg_pitch = 0;
      c->cg_ps1 = 0;
    }
  }
  // the tiled phases do not visit the last boundary row / column of non-periodic axes (launch_cg3d): p, v, s, t are 0
  // there by definition and have to be 0 in every buffer the phases write into (and in the pad cells of pitched rows)
  for (T* q : {p[1], v[1], s, t}) PA_HIP(c, hipMemsetAsync(q, 0, fb, c->stream));
  if (!c->cg_pitch && (rc = cg_residual_init<T>(c, E, xv, rhs, r0, r, (T*)nullptr, (T*)nullptr, part))) return rc;
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
  const bool bicg_static = pa_bc_is_static(c);
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
  // the next direction formed by k_bicg_x (one array pass less per iteration, one haloed input instead of three in the
  // p / v phase); option "bicg_pfold" 0: every iteration through the p / v phase, as before round 3
  const bool pfold = c->bicg_pfold != 0;
  bool pgiven = false;
  c->fold_a_n = c->fold_b_n = c->fold_b_nsh = 0;
  if (c->coord == PA_COORD_RZ && pfold && c->fastpath) {
    // Axisymmetric mesh: the only tiled form of the p / v phase is the one that takes p' as given (k_cg2d<..., RZ>, phase
    // 8).  The first iteration has p = v = 0, so its p' = r + beta (0 - omega 0) IS r, bit for bit (linalg.py:189-217):
    // hand phase 8 a copy of r and every iteration -- the first included -- runs on the marching kernel.
    c->plan_only = 1;
    const int v_ok = pa_tile3d_bicg_v<T>(c, E, vec_of(p[0]), (const T*)r0, v[1], reg0);
    c->plan_only = 0;
    (void)hipGetLastError();
    if (v_ok > 0) {
      PA_HIP(c, hipMemcpyAsync(p[0], r, fb, hipMemcpyDeviceToDevice, c->stream));
      pgiven = true;
    }
  }
  if (c->cg_pitch && c->coord == PA_COORD_RZ && !pgiven) { pa_set_err(c, "pitched BiCGSTAB on an axisymmetric mesh needs the marching v phase"); return PA_E_STATE; }
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
    Vec<T> rv = vec_of(r), pv = vec_of(p[cur]), vv = vec_of(v[cur]);
    c->fold_b_n = pend3;          // phase 5 closes the previous iteration (and swaps the scalar slots)
    c->fold_b_part = reg2;
    // p' of this iteration: formed by the p / v phase into p[cur ^ 1] -- or already there, in p[cur], left by the
    // previous iteration's k_bicg_x (`pgiven`; tiled kernels only), and the phase is v' = A p' alone
    T* const p_it = pgiven ? p[cur] : p[cur ^ 1];
    int used = pgiven ? pa_tile3d_bicg_v<T>(c, E, pv, (const T*)r0, v[cur ^ 1], reg0)
                      : pa_tile3d_bicg_pv<T>(c, E, rv, pv, vv, (const T*)r0, p[cur ^ 1], v[cur ^ 1], reg0);
    if (used < 0) return used;
    if (pgiven && used == 0) { pa_set_err(c, "pa_bicgstab: the tiled v phase declined in the middle of a solve"); return PA_E_STATE; }
    if (c->cg_pitch && used == 0) { pa_set_err(c, "pitched BiCGSTAB: the tiled p / v phase declined"); return PA_E_STATE; }
    const bool pnext = pfold && used > 0;   // the tiled kernels took this iteration: they take the next one
    if (used > 0) {
      pend3 = 0;
    } else {
      flush3();
      hipLaunchKernelGGL(k_bicg_pv<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, G, E, c->sc, rv, pv, vv, (const T*)r0,
                         p[cur ^ 1], v[cur ^ 1], reg0);
      used = nblk;
    }
    if (c->x_old_out)   // after the p / v phase: its prologue has decided whether this iteration still runs
      hipLaunchKernelGGL(k_copy_guarded<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, c->sc, (const T*)x,
                         (T*)c->x_old_out, G.ncell);
    int pend0 = (fold && used <= PA_MAX_GRID) ? used : 0;
    if (!pend0) hipLaunchKernelGGL(k_bicg_post<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, reg0, used, 0);
    Vec<T> vnv = vec_of(v[cur ^ 1]);
    c->fold_a_n = pend0;          // phase 6 computes alpha itself
    // (option "bicg_srv", default on: the tiled phase stores t alone and k_bicg_x re-forms s from r and v')
    int used2 = pa_tile3d_bicg_st<T>(c, E, rv, vnv, (const T*)r0, c->bicg_srv ? (T*)nullptr : s, t, reg1);
    const bool srv = c->bicg_srv && used2 > 0;
    c->fold_a_n = 0;
    if (used2 < 0) return used2;
    if (c->cg_pitch && used2 == 0) { pa_set_err(c, "pitched BiCGSTAB: the tiled s / t phase declined"); return PA_E_STATE; }
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
    {
      constexpr int XV = 16 / (int)sizeof(T);
      const bool vec = G.ncell % XV == 0 && ((((uintptr_t)x | (uintptr_t)p[cur ^ 1] | (uintptr_t)s | (uintptr_t)t | (uintptr_t)r) & 15) == 0);
#define PA_BICG_X(VV, PP, SS, ...)                                                                                          \
      hipLaunchKernelGGL((k_bicg_x<T, VV, PP, SS>), dim3(nblk), dim3(PA_BLOCK), 0, c->stream, G, c->sc, x, (const T*)p_it, \
                         (const T*)s, (const T*)t, r, reg2, (const double*)reg1, pend12, c->sc, (const T*)v[cur ^ 1],      \
                         pnext ? p[cur ^ 1] : (T*)nullptr, ##__VA_ARGS__)
      if (c->cg_pitch) {
        if (srv) PA_BICG_X(XV, true, true, c->cg_ps1); else PA_BICG_X(XV, true, false, c->cg_ps1);
      } else if (vec) {
        if (srv) PA_BICG_X(XV, false, true); else PA_BICG_X(XV, false, false);
      } else {
        if (srv) PA_BICG_X(1, false, true); else PA_BICG_X(1, false, false);
      }
#undef PA_BICG_X
      pgiven = pnext;   // (p[cur ^ 1] is p[cur] of the next iteration; in place when p' came from the p / v phase)
    }
    // Dirichlet faces only: the fill of pa_bicg's set-up stands -- p and s are +-0 on every boundary node, so the x / r
    // update leaves x there as it is (alpha, omega are finite by pa_nan_to_num) and a fill would rewrite the same values
    // (the CG loop skips it the same way): one launch less per iteration, 26 us of 2.9 ms at 512^3, 6 of 40 us at 64^3
    if (!bicg_static && (rc = pa_bc_apply_auto<T>(c, x, true))) return rc;
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

int pa_solver_keep_old(pa_ctx* c, void* x_old) {
  if (!c) return PA_E_ARG;
  if (x_old && c->solver_live) { pa_set_err(c, "pa_solver_keep_old during a solve"); return PA_E_STATE; }
  c->x_old_out = x_old;
  return PA_OK;
}

int pa_cg(pa_ctx* c, void* x, const void* rhs, double tol, int64_t max_it, pa_report* out) {
  if (!c || !c->grid_set || !c->eq_set) { if (c) pa_set_err(c, "pa_cg: grid/equation not set"); return PA_E_STATE; }
  if (int rc0 = pa_check_eq_applicable(c)) return rc0;
  if (!out) return PA_E_ARG;
  if (c->slab) { pa_set_err(c, "pa_cg is the single-GPU loop; use the stepwise API on a slab"); return PA_E_STATE; }
  PA_HIP(c, hipSetDevice(c->device));
  const int rc = c->dtype == PA_F64 ? cg_run_t<double>(c, (double*)x, (const double*)rhs, tol, max_it, out)
                                    : cg_run_t<float>(c, (float*)x, (const float*)rhs, tol, max_it, out);
  c->solver_live = 0;   // also on the error paths: a failed one-shot solve must not lock the BC / equation state
  pa_place_end(c, 0);
  return rc;
}

int pa_bicgstab(pa_ctx* c, void* x, const void* rhs, double tol, int64_t max_it, pa_report* out) {
  if (!c || !c->grid_set || !c->eq_set) { if (c) pa_set_err(c, "pa_bicgstab: grid/equation not set"); return PA_E_STATE; }
  if (int rc0 = pa_check_eq_applicable(c)) return rc0;
  if (!out) return PA_E_ARG;
  PA_HIP(c, hipSetDevice(c->device));
  return c->dtype == PA_F64 ? bicg_run_t<double>(c, (double*)x, (const double*)rhs, tol, max_it, out)
                            : bicg_run_t<float>(c, (float*)x, (const float*)rhs, tol, max_it, out);
}

int pa_jacobi(pa_ctx* c, void* x, const void* rhs, double tol, int64_t max_it, double omega, pa_report* out) {
  if (!c || !c->grid_set || !c->eq_set) { if (c) pa_set_err(c, "pa_jacobi: grid/equation not set"); return PA_E_STATE; }
  if (int rc0 = pa_check_eq_applicable(c)) return rc0;
  if (!out) return PA_E_ARG;
  PA_HIP(c, hipSetDevice(c->device));
  return c->dtype == PA_F64 ? jacobi_run_t<double>(c, (double*)x, (const double*)rhs, tol, max_it, omega, out)
                            : jacobi_run_t<float>(c, (float*)x, (const float*)rhs, tol, max_it, omega, out);
}

}  // extern "C"

// ============================================================================
//  stepwise CG (bench.py, slab-decomposed driver)
// ============================================================================
// rows[0..2] = partial rows this rank's tiled phase A / phase B / BC fill write per iteration (all 0:
// the folded slab sequence does not apply here -- generic kernels, or too many rows)
template <typename T>
static int cg_fold_plan_t(pa_ctx* c, int64_t* rows) {
  rows[0] = rows[1] = rows[2] = 0;
  DevEq<T> E;
  pa_build_eq<T>(c, c->nterms, c->terms, E);
  T* r = (T*)c->scr[SCR_R];
  T* d0 = (T*)c->scr[SCR_D0];
  T* d1 = (T*)c->scr[SCR_D1];
  double* part = (double*)c->scr[SCR_PART];
  c->plan_only = 1;
  const int na = pa_cg3d_phase_a<T>(c, E, cg_vec<T>(c, r, 0), cg_vec<T>(c, d0, 1), d1, part);
  const int nb = pa_cg3d_phase_b<T>(c, E, cg_vec<T>(c, d1, 1), (T*)c->cg_x, r, part);
  c->plan_only = 0;
  (void)hipGetLastError();
  const int ns = c->bc_static ? 0 : pa_bc_shell_rows(c);
  if (na <= 0 || nb <= 0 || na > PA_MAX_GRID || nb > PA_MAX_GRID || ns > 3 * PA_MAX_GRID) return PA_OK;
  rows[0] = na; rows[1] = nb; rows[2] = ns;
  return PA_OK;
}

extern "C" {

int pa_cg_begin(pa_ctx* c, void* x, const void* rhs, double tol, int64_t max_it) {
  if (!c || !c->grid_set || !c->eq_set) { if (c) pa_set_err(c, "pa_cg_begin: grid/equation not set"); return PA_E_STATE; }
  if (int rc0 = pa_check_eq_applicable(c)) return rc0;
  PA_HIP(c, hipSetDevice(c->device));
  return c->dtype == PA_F64 ? cg_begin_t<double>(c, (double*)x, (const double*)rhs, tol, max_it)
                            : cg_begin_t<float>(c, (float*)x, (const float*)rhs, tol, max_it);
}

int pa_cg_phase_a(pa_ctx* c) {
  if (!c || !c->solver_live) { if (c) pa_set_err(c, "pa_cg_phase_a without pa_cg_begin"); return PA_E_STATE; }
  PA_HIP(c, hipSetDevice(c->device));
  const int st = c->slab ? 0 : 2;
  return c->dtype == PA_F64 ? pa_cg_phase_a_t<double>(c, st) : pa_cg_phase_a_t<float>(c, st);
}

int pa_cg_phase_b(pa_ctx* c) {
  if (!c || !c->solver_live) { if (c) pa_set_err(c, "pa_cg_phase_b without pa_cg_begin"); return PA_E_STATE; }
  PA_HIP(c, hipSetDevice(c->device));
  if (c->slab && !c->slab_fold_live) {  // alpha from the all-reduced sum d.Ad
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
  PA_HIP(c, hipSetDevice(c->device));
  if (!c->slab) return PA_OK;  // done inside phase_b
  return c->dtype == PA_F64 ? pa_cg_bc_t<double>(c) : pa_cg_bc_t<float>(c);
}

int pa_cg_finish_iter(pa_ctx* c) {
  if (!c || !c->solver_live) { if (c) pa_set_err(c, "pa_cg_finish_iter without pa_cg_begin"); return PA_E_STATE; }
  PA_HIP(c, hipSetDevice(c->device));
  if (!c->slab || c->slab_fold_live) return PA_OK;  // logic already ran inside phase_b / runs in the next prologue
  if (c->dtype == PA_F64)
    hipLaunchKernelGGL(k_cg_post_b<double>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, (const double*)nullptr, 0,
                       (const double*)nullptr, 0, pa_sums(c), 1);
  else
    hipLaunchKernelGGL(k_cg_post_b<float>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, (const double*)nullptr, 0,
                       (const double*)nullptr, 0, pa_sums(c), 1);
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

int pa_cg_fold_plan(pa_ctx* c, int64_t* rows) {
  if (!c || !rows) return PA_E_ARG;
  if (!c->solver_live || !c->slab) { pa_set_err(c, "pa_cg_fold_plan needs a live slab solve (pa_slab_set, pa_cg_begin)"); return PA_E_STATE; }
  return c->dtype == PA_F64 ? cg_fold_plan_t<double>(c, rows) : cg_fold_plan_t<float>(c, rows);
}

int pa_cg_fold_set(pa_ctx* c, const int64_t* rows) {
  if (!c) return PA_E_ARG;
  if (!c->solver_live || !c->slab) { pa_set_err(c, "pa_cg_fold_set needs a live slab solve"); return PA_E_STATE; }
  c->slab_fold = 0;
  if (!rows || rows[0] <= 0 || rows[1] <= 0 || rows[2] < 0) return PA_OK;   // stepwise sequence
  if (rows[0] > PA_MAX_GRID || rows[1] > PA_MAX_GRID || rows[2] > 3 * PA_MAX_GRID) {
    pa_set_err(c, "pa_cg_fold_set: row counts beyond one resident wave of workgroups");
    return PA_E_ARG;
  }
  int64_t mine[3];
  if (int rc = pa_cg_fold_plan(c, mine)) return rc;
  if (mine[0] <= 0 || mine[0] > rows[0] || mine[1] > rows[1] || mine[2] > rows[2]) {
    pa_set_err(c, "pa_cg_fold_set: agreed rows (%lld %lld %lld) below this rank's (%lld %lld %lld)", (long long)rows[0],
               (long long)rows[1], (long long)rows[2], (long long)mine[0], (long long)mine[1], (long long)mine[2]);
    return PA_E_ARG;
  }
  const size_t tot = (size_t)rows[0] + 2 * (size_t)rows[1] + (size_t)rows[2];
  PA_HIP(c, hipSetDevice(c->device));
  if (tot > c->rows_cap) {
    if (c->rows_buf[0]) (void)hipFree(c->rows_buf[0]);
    if (c->rows_buf[1]) (void)hipFree(c->rows_buf[1]);
    c->rows_buf[0] = c->rows_buf[1] = nullptr;
    c->rows_cap = 0;
    PA_HIP(c, hipMalloc((void**)&c->rows_buf[0], tot * sizeof(double)));
    PA_HIP(c, hipMalloc((void**)&c->rows_buf[1], tot * sizeof(double)));
    c->rows_cap = tot;
  }
  // Rows beyond this rank's own grids are never written: they must be (and stay) zero in the send buffer,
  // hence the all-reduce out of place.  A rank whose counts ARE the agreed ones rewrites every row in every
  // iteration and reduces in place (a local choice: RCCL does not care whether send == recv on a rank).
  c->rows_send = c->rows_buf[0];
  c->rows_recv = (mine[0] == rows[0] && mine[1] == rows[1] && mine[2] == rows[2]) ? c->rows_buf[0] : c->rows_buf[1];
  PA_HIP(c, hipMemsetAsync(c->rows_buf[0], 0, tot * sizeof(double), c->stream));
  PA_HIP(c, hipMemsetAsync(c->rows_buf[1], 0, tot * sizeof(double), c->stream));
  for (int q = 0; q < 3; ++q) c->fold_rows[q] = (int)rows[q];
  c->slab_fold = 1;
  return PA_OK;
}

int pa_cg_iterate(pa_ctx* c, int64_t n) {
  if (!c || !c->solver_live) { if (c) pa_set_err(c, "pa_cg_iterate without pa_cg_begin"); return PA_E_STATE; }
  PA_HIP(c, hipSetDevice(c->device));
  if (c->slab) { pa_set_err(c, "pa_cg_iterate is single-rank; drive the phases on a slab"); return PA_E_STATE; }
  c->in_iterate = 1;
  int rc = PA_OK;
  for (int64_t q = 0; q < n && !rc; ++q) {
    if ((rc = pa_place_tick(c))) break;
    rc = c->dtype == PA_F64 ? pa_cg_phase_a_t<double>(c, 2) : pa_cg_phase_a_t<float>(c, 2);
    if (!rc) rc = c->dtype == PA_F64 ? pa_cg_phase_b_t<double>(c, 2) : pa_cg_phase_b_t<float>(c, 2);
  }
  c->in_iterate = 0;
  if (!rc) rc = pa_place_batch_end(c);
  if (c->dtype == PA_F64) cg_flush_fold<double>(c); else cg_flush_fold<float>(c);  // the last iteration's stop test
  return rc;
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

int pa_scalars_read(pa_ctx* c, double* out) {
  if (!c || !out) return PA_E_ARG;
  if (c->solver_live) {   // a stepwise solve: fetch the live state; after pa_cg / pa_bicgstab / pa_jacobi the mirror is current
    int rc = read_scalars(c);
    if (rc) return rc;
  }
  const SolverScalars& h = *c->h_sc;
  const double v[PA_NSCALAR] = {h.alpha, h.beta, h.rr, h.rr_old, h.dAd, h.tol, h.rho, h.omega, h.rho_next, h.r0v,
                                h.ts, h.tt, h.r0t, (double)h.itr, 0.0, 0.0};
  for (int q = 0; q < PA_NSCALAR; ++q) out[q] = v[q];
  return PA_OK;
}

int pa_cg_abort(pa_ctx* c) {   // drop a stepwise solve without reading it back (error paths of a host driver)
  if (!c) return PA_E_ARG;
  c->solver_live = 0;
  c->in_iterate = c->slab_fold_live = 0;
  c->fold_a_n = c->fold_b_n = c->fold_b_nsh = 0;
  c->fold_b_shell = nullptr;
  pa_place_end(c, 0);   // (no hipFree here: it would wait for a stream that may never drain)
  return PA_OK;
}

int pa_cg_end(pa_ctx* c, pa_report* out) {
  if (!c || !c->solver_live) { if (c) pa_set_err(c, "pa_cg_end without pa_cg_begin"); return PA_E_STATE; }
  PA_HIP(c, hipSetDevice(c->device));
  int rc = out ? pa_report_read(c, out) : PA_OK;
  c->solver_live = 0;
  pa_place_end(c, out ? 1 : 0);   // (pa_report_read has waited for the stream)
  if (rc) return rc;
  return (out && out->status) ? PA_E_NONFINITE : PA_OK;
}

}  // extern "C"


// ============================================================================
//  stepwise BiCGSTAB on a slab (linalg.py:162-279 split at its reductions and exchanges)
// ============================================================================
// Config 3 is periodic: CG never meets the reference's stop test there, BiCGSTAB is the solver that
// converges -- so it has to exist on slabs too.  Per iteration, with the planes a rank needs from its axis-0
// neighbours:
//   pv      p' = r + beta (p - omega v) -- on the ghost planes too, from the ghost planes of r, p, v with the owner's
//           recurrence bit for bit, so p is never exchanged -- ; v' = A p' on S ; local sum r0.v'   -> [all-reduce 1]
//                                                                                 -> [exchange the boundary planes of v']
//   st      alpha ; s = r - alpha v' (ghost planes from those of r and v') ; t = A s on S ;
//           local sums |s|^2, t.s, t.t, r0.t                                       -> [all-reduce 4]
//   x       stop test 1, omega, rho' ; x += alpha p' + omega s ; r = s - omega t   -> [exchange r planes (+ periodic x planes)]
//   bc      BC fill of x ; local sum |r|^2                                         -> [all-reduce 1]
//   finish  stop test 2, beta, rho <- rho'
// Two plane exchanges and three small all-reduces per iteration.  The sums travel in the caller's PA_NSUM buffer:
// [0] r0.v', [1] |s|^2 (and r0.r0 of the start), [2] t.s, [3] t.t, [4] r0.t, [5] |r|^2.  Kernels: the tiled phases 5 / 6
// where they apply (they take ghost planes through Vec<T>), else the generic ones; the next direction is NOT folded into
// the x / r update here (its ghost planes would need the new residual's, which is still on the wire).
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_rows_to_sums(const SolverScalars* __restrict__ sc, const double* __restrict__ partials,
                                                            int nblk, int ncol, double* __restrict__ sums, int off, int guarded) {
  __shared__ double sm[PA_BLOCK / 64];
  if (guarded && sc->done) return;
  for (int q = 0; q < ncol; ++q) {
    const double v = pa_reduce_partials(partials, nblk, ncol, q, sm);
    if (threadIdx.x == 0) sums[off + q] = v;
  }
}

// the scalar steps of k_bicg_post, from all-reduced sums instead of partial rows.  stage 10: start (rho' = r0.r0,
// first beta); 0: alpha; 12: stop test 1, omega, rho'; 3: stop test 2, next beta
template <typename T>
__global__ void k_bicg_logic(SolverScalars* sc, const double* __restrict__ sums, int stage) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (stage == 10) {
    sc->rr = (double)(T)sums[1];
    sc->rho_next = sc->rr;
    sc->tol = (double)(T)sqrt((T)sc->rr);
    T b = (T)sc->rho_next / (T)1.0;
    b = b * (T)1.0;
    b = b / (T)1.0;
    sc->beta = (double)b;
    sc->rho = sc->rho_next;
    sc->done = 0;   // `while not finished`: at least one iteration
    return;
  }
  if (sc->done) return;
  if (stage == 0) {
    sc->itr += 1;
    T r0v = (T)sums[0];
    T rho = (T)sc->rho;
    sc->alpha = pa_nan_to_num<T>(rho / r0v);
  } else if (stage == 12) {
    T tol = (T)sqrt(sums[1]);
    sc->tol = (double)tol;
    if (isnan(tol) || isinf(tol)) { sc->err = 1; sc->done = 1; return; }
    sc->finished_early = (sc->tol <= sc->tolerance) ? 1 : 0;
    if (!sc->finished_early) {
      T om = (T)pa_nan_to_num<T>((T)sums[2] / (T)sums[3]);
      sc->omega = (double)om;
      T rn = -om;
      rn = rn * (T)sums[4];
      sc->rho_next = (double)rn;
    }
  } else {
    if (sc->finished_early) { sc->done = 1; return; }
    T tol = (T)sqrt(sums[5]);
    sc->tol = (double)tol;
    if (isnan(tol) || isinf(tol)) { sc->err = 1; sc->done = 1; return; }
    if (sc->tol <= sc->tolerance) sc->done = 1;
    if (sc->itr >= sc->max_it) sc->done = 1;
    T b = (T)sc->rho_next / (T)sc->rho;
    b = b * (T)sc->alpha;
    b = b / (T)sc->omega;
    sc->beta = (double)b;
    sc->rho = sc->rho_next;
  }
}

// ghost planes of p' = r + beta (p - omega v): the owner's recurrence (BicgPAcc::at / the combine of phase 5)
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_ghost_p(const SolverScalars* __restrict__ sc, int64_t n,
                                                       const T* __restrict__ r_lo, const T* __restrict__ r_hi,
                                                       const T* __restrict__ p_lo, const T* __restrict__ p_hi,
                                                       const T* __restrict__ v_lo, const T* __restrict__ v_hi,
                                                       T* __restrict__ o_lo, T* __restrict__ o_hi) {
  if (sc->done) return;
  const T beta = (T)sc->beta, omega = (T)sc->omega;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
    if (r_lo) { T t = omega * v_lo[q]; t = p_lo[q] - t; t = beta * t; o_lo[q] = r_lo[q] + t; }
    if (r_hi) { T t = omega * v_hi[q]; t = p_hi[q] - t; t = beta * t; o_hi[q] = r_hi[q] + t; }
  }
}

// ghost planes of s = r - alpha v' (generic kernels only: the tiled phase 6 forms them on load)
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_ghost_s(const SolverScalars* __restrict__ sc, int64_t n,
                                                       const T* __restrict__ r_lo, const T* __restrict__ r_hi,
                                                       const T* __restrict__ v_lo, const T* __restrict__ v_hi,
                                                       T* __restrict__ o_lo, T* __restrict__ o_hi) {
  if (sc->done) return;
  const T alpha = (T)sc->alpha;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
    if (r_lo) { T av = alpha * v_lo[q]; o_lo[q] = r_lo[q] - av; }
    if (r_hi) { T av = alpha * v_hi[q]; o_hi[q] = r_hi[q] - av; }
  }
}

namespace {

// a field of the slab solver with its ghost planes; a physical (non-periodic) end has none: no result uses that plane
// (the end plane is a boundary node), the field's own end plane stands in so that speculative loads stay in valid memory
template <typename T>
Vec<T> slab_vec(const pa_ctx* c, const T* p, const void* glo, const void* ghi) {
  Vec<T> v;
  v.p = p;
  v.glo = glo ? (const T*)glo : p;
  v.ghi = ghi ? (const T*)ghi : p + (c->G.n0 - 1) * c->G.s0;
  return v;
}

template <typename T>
int bicg_slab_begin_t(pa_ctx* c, T* x, const T* rhs, double tol, int64_t max_it) {
  const DevGeom& G = c->G;
  const size_t fb = (size_t)G.ncell * sizeof(T), pb = (size_t)G.s0 * sizeof(T);
  const int nblk = pa_grid_blocks(G.ncell);
  int rc;
  c->cg_pitch = 0;
  c->cg_ps1 = 0;
  const int ids[] = {SCR_R, SCR_D0, SCR_D1, SCR_R0, SCR_V0, SCR_V1, SCR_S, SCR_TT};
  for (int id : ids)
    if ((rc = pa_scratch(c, &c->scr[id], &c->cap[id], fb))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_PART], &c->cap[SCR_PART], (size_t)PA_MAX_PARTIALS * 6 * sizeof(double)))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_GHOST], &c->cap[SCR_GHOST], 6 * pb))) return rc;   // p ghosts (lo / hi x ping / pong), s ghosts
  if ((rc = init_scalars(c, tol, max_it))) return rc;
  DevEq<T> E;
  pa_build_eq<T>(c, c->nterms, c->terms, E);
  T* r = (T*)c->scr[SCR_R];
  T* r0 = (T*)c->scr[SCR_R0];
  // (the driver has filled the BCs and exchanged the ghost planes of x: linalg.py:181)
  Vec<T> xv = pa_vec_self<T>(c, x);
  xv.glo = (const T*)c->x_glo;
  xv.ghi = (const T*)c->x_ghi;
  for (int id : {SCR_D0, SCR_D1, SCR_V0, SCR_V1, SCR_S, SCR_TT}) PA_HIP(c, hipMemsetAsync(c->scr[id], 0, fb, c->stream));
  PA_HIP(c, hipMemsetAsync(c->scr[SCR_GHOST], 0, 6 * pb, c->stream));
  // r0 = r = b - A x on S ; local sum r0.r0 ; first / last owned plane of r for the neighbours
  if ((rc = cg_residual_init<T>(c, E, xv, rhs, r0, r, (T*)c->r_send_lo, (T*)c->r_send_hi, (double*)c->scr[SCR_PART]))) return rc;
  hipLaunchKernelGGL(k_rows_to_sums<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, (const double*)c->scr[SCR_PART], nblk, 1,
                     pa_sums(c), 1, 0);
  c->cg_x = x;
  c->cur = 0;
  c->solver_live = 2;   // (2: the stepwise BiCGSTAB)
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

template <typename T>
int bicg_slab_pv_t(pa_ctx* c) {
  const DevGeom& G = c->G;
  const int nblk = pa_grid_blocks(G.ncell);
  const size_t pl = (size_t)G.s0;
  DevEq<T> E;
  pa_build_eq<T>(c, c->nterms, c->terms, E);
  T* r = (T*)c->scr[SCR_R];
  T* r0 = (T*)c->scr[SCR_R0];
  T* p[2] = {(T*)c->scr[SCR_D0], (T*)c->scr[SCR_D1]};
  T* v[2] = {(T*)c->scr[SCR_V0], (T*)c->scr[SCR_V1]};
  T* g = (T*)c->scr[SCR_GHOST];
  T* pg_lo[2] = {g, g + 2 * pl};
  T* pg_hi[2] = {g + pl, g + 3 * pl};
  double* part = (double*)c->scr[SCR_PART];
  const int cur = c->cur;
  const bool lo = c->r_recv_lo != nullptr, hi = c->r_recv_hi != nullptr;
  if ((lo && !c->v_recv_lo) || (hi && !c->v_recv_hi)) { pa_set_err(c, "pa_bicg_pv: a neighbour without a receive plane for v (pa_slab_set_v)"); return PA_E_STATE; }
  Vec<T> rv = slab_vec<T>(c, r, c->r_recv_lo, c->r_recv_hi);
  Vec<T> pv = slab_vec<T>(c, p[cur], lo ? pg_lo[cur] : nullptr, hi ? pg_hi[cur] : nullptr);
  Vec<T> vv = slab_vec<T>(c, v[cur], c->v_recv_lo, c->v_recv_hi);
  if (lo || hi)   // the ghost planes of p' for the NEXT iteration (this one forms them on load)
    hipLaunchKernelGGL(k_ghost_p<T>, dim3(pa_grid_blocks(G.s0)), dim3(PA_BLOCK), 0, c->stream, c->sc, G.s0,
                       lo ? (const T*)c->r_recv_lo : nullptr, hi ? (const T*)c->r_recv_hi : nullptr, (const T*)pg_lo[cur],
                       (const T*)pg_hi[cur], (const T*)c->v_recv_lo, (const T*)c->v_recv_hi, pg_lo[cur ^ 1], pg_hi[cur ^ 1]);
  c->fold_a_n = c->fold_b_n = 0;
  int used = pa_tile3d_bicg_pv<T>(c, E, rv, pv, vv, (const T*)r0, p[cur ^ 1], v[cur ^ 1], part);
  if (used < 0) return used;
  if (used == 0) {
    hipLaunchKernelGGL(k_bicg_pv<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, G, E, c->sc, rv, pv, vv, (const T*)r0,
                       p[cur ^ 1], v[cur ^ 1], part);
    used = nblk;
  }
  hipLaunchKernelGGL(k_rows_to_sums<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, (const double*)part, used, 1, pa_sums(c), 0, 1);
  if (c->v_send_lo || c->v_send_hi)
    hipLaunchKernelGGL(k_pack_planes<T>, dim3(pa_grid_blocks(G.s0)), dim3(PA_BLOCK), 0, c->stream, c->sc, G.s0,
                       (const T*)v[cur ^ 1], (T*)c->v_send_lo, (const T*)v[cur ^ 1] + (G.n0 - 1) * G.s0, (T*)c->v_send_hi,
                       (const T*)nullptr, (T*)nullptr);
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

template <typename T>
int bicg_slab_st_t(pa_ctx* c) {
  const DevGeom& G = c->G;
  const int nblk = pa_grid_blocks(G.ncell);
  const size_t pl = (size_t)G.s0;
  DevEq<T> E;
  pa_build_eq<T>(c, c->nterms, c->terms, E);
  T* r = (T*)c->scr[SCR_R];
  T* r0 = (T*)c->scr[SCR_R0];
  T* vn = (T*)c->scr[c->cur ? SCR_V0 : SCR_V1];   // v' of this iteration
  T* s = (T*)c->scr[SCR_S];
  T* t = (T*)c->scr[SCR_TT];
  T* g = (T*)c->scr[SCR_GHOST];
  T* sg_lo = g + 4 * pl;
  T* sg_hi = g + 5 * pl;
  double* part = (double*)c->scr[SCR_PART] + (size_t)PA_MAX_PARTIALS;
  hipLaunchKernelGGL(k_bicg_logic<T>, dim3(1), dim3(1), 0, c->stream, c->sc, (const double*)pa_sums(c), 0);   // alpha, itr
  Vec<T> rv = slab_vec<T>(c, r, c->r_recv_lo, c->r_recv_hi);
  Vec<T> vv = slab_vec<T>(c, vn, c->v_recv_lo, c->v_recv_hi);
  c->fold_a_n = 0;
  // (as on one GPU: the tiled phase stores t alone and the x / r step re-forms s from r and v', option "bicg_srv")
  int used = pa_tile3d_bicg_st<T>(c, E, rv, vv, (const T*)r0, c->bicg_srv ? (T*)nullptr : s, t, part);
  if (used < 0) return used;
  c->bicg_s_stored = (used > 0 && c->bicg_srv) ? 0 : 1;
  if (used > 0) {
    hipLaunchKernelGGL(k_rows_to_sums<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, (const double*)part, used, 4, pa_sums(c), 1, 1);
  } else {
    // generic kernels: s everywhere, its ghost planes, t = A s on S.  (Unlike the one-GPU loop, t is formed even when
    // the first stop test is about to end the solve: the test needs the all-reduced |s|^2, which comes after this call.)
    hipLaunchKernelGGL(k_bicg_s<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, G, c->sc, (const T*)r, (const T*)vn, s, part);
    hipLaunchKernelGGL(k_rows_to_sums<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, (const double*)part, nblk, 1, pa_sums(c), 1, 1);
    const bool lo = c->r_recv_lo != nullptr, hi = c->r_recv_hi != nullptr;
    if (lo || hi)
      hipLaunchKernelGGL(k_ghost_s<T>, dim3(pa_grid_blocks(G.s0)), dim3(PA_BLOCK), 0, c->stream, c->sc, G.s0,
                         lo ? (const T*)c->r_recv_lo : nullptr, hi ? (const T*)c->r_recv_hi : nullptr, (const T*)c->v_recv_lo,
                         (const T*)c->v_recv_hi, sg_lo, sg_hi);
    Vec<T> sv = slab_vec<T>(c, s, lo ? sg_lo : nullptr, hi ? sg_hi : nullptr);
    hipLaunchKernelGGL(k_bicg_t<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, G, E, c->sc, sv, (const T*)r0, t, part);
    hipLaunchKernelGGL(k_rows_to_sums<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, (const double*)part, nblk, 3, pa_sums(c), 2, 1);
  }
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

template <typename T>
int bicg_slab_x_t(pa_ctx* c) {
  const DevGeom& G = c->G;
  const int nblk = pa_grid_blocks(G.ncell);
  T* x = (T*)c->cg_x;
  T* r = (T*)c->scr[SCR_R];
  T* pn = (T*)c->scr[c->cur ? SCR_D0 : SCR_D1];   // p' of this iteration
  T* s = (T*)c->scr[SCR_S];
  T* t = (T*)c->scr[SCR_TT];
  double* part = (double*)c->scr[SCR_PART] + 5 * (size_t)PA_MAX_PARTIALS;
  hipLaunchKernelGGL(k_bicg_logic<T>, dim3(1), dim3(1), 0, c->stream, c->sc, (const double*)pa_sums(c), 12);   // stop test 1, omega, rho'
  constexpr int XV = 16 / (int)sizeof(T);
  const bool vec = G.ncell % XV == 0 && ((((uintptr_t)x | (uintptr_t)pn | (uintptr_t)s | (uintptr_t)t | (uintptr_t)r) & 15) == 0);
  const T* vn = (const T*)c->scr[c->cur ? SCR_V0 : SCR_V1];   // v' of this iteration (s = r - alpha v' when s was not stored)
#define PA_BICG_XS(VV, SS)                                                                                                        \
  hipLaunchKernelGGL((k_bicg_x<T, VV, false, SS>), dim3(nblk), dim3(PA_BLOCK), 0, c->stream, G, c->sc, x, (const T*)pn, (const T*)s, \
                     (const T*)t, r, part, (const double*)nullptr, 0, c->sc, SS ? vn : (const T*)nullptr, (T*)nullptr)
  if (vec) {
    if (c->bicg_s_stored) PA_BICG_XS(XV, false); else PA_BICG_XS(XV, true);
  } else {
    if (c->bicg_s_stored) PA_BICG_XS(1, false); else PA_BICG_XS(1, true);
  }
#undef PA_BICG_XS
  // what the neighbours need next: the first / last owned plane of the new residual and, on the end ranks of a
  // periodic ring, the x planes the other end's BC fill reads (packed behind them by the driver's buffer layout)
  if (c->r_send_lo || c->r_send_hi)
    hipLaunchKernelGGL(k_pack_planes<T>, dim3(pa_grid_blocks(G.s0)), dim3(PA_BLOCK), 0, c->stream, c->sc, G.s0,
                       (const T*)r, (T*)c->r_send_lo, (const T*)r + (G.n0 - 1) * G.s0, (T*)c->r_send_hi, (const T*)nullptr, (T*)nullptr);
  if (c->x_pack_lo1 || c->x_pack_hi0 || c->x_pack_hi1) {
    const T* xr = (const T*)x;
    hipLaunchKernelGGL(k_pack_planes<T>, dim3(pa_grid_blocks(G.s0)), dim3(PA_BLOCK), 0, c->stream, c->sc, G.s0,
                       xr + 1 * G.s0, (T*)c->x_pack_lo1, xr + (G.n0 - 1) * G.s0, (T*)c->x_pack_hi0,
                       xr + (G.n0 - 2) * G.s0, (T*)c->x_pack_hi1);
  }
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

template <typename T>
int bicg_slab_bc_t(pa_ctx* c) {
  const int nblk = pa_grid_blocks(c->G.ncell);
  int rc = pa_bc_is_static(c) ? PA_OK : pa_bc_apply_auto<T>(c, (T*)c->cg_x, true);   // (Dirichlet faces only: bicg_run_t)
  if (rc) return rc;
  hipLaunchKernelGGL(k_rows_to_sums<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc,
                     (const double*)c->scr[SCR_PART] + 5 * (size_t)PA_MAX_PARTIALS, nblk, 1, pa_sums(c), 5, 1);
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

}  // namespace

extern "C" {

int pa_slab_set_v(pa_ctx* c, void* v_send_lo, void* v_send_hi, const void* v_recv_lo, const void* v_recv_hi) {
  if (!c) return PA_E_ARG;
  if (c->solver_live) { pa_set_err(c, "pa_slab_set_v during a solve"); return PA_E_STATE; }
  c->v_send_lo = v_send_lo; c->v_send_hi = v_send_hi;
  c->v_recv_lo = v_recv_lo; c->v_recv_hi = v_recv_hi;
  return PA_OK;
}

#define PA_BICG_LIVE(c, what)                                                                                         \
  if (!(c) || (c)->solver_live != 2) { if (c) pa_set_err((c), what " without pa_bicg_begin"); return PA_E_STATE; }     \
  PA_HIP((c), hipSetDevice((c)->device));

int pa_bicg_begin(pa_ctx* c, void* x, const void* rhs, double tol, int64_t max_it) {
  if (!c || !c->grid_set || !c->eq_set) { if (c) pa_set_err(c, "pa_bicg_begin: grid/equation not set"); return PA_E_STATE; }
  if (int rc0 = pa_check_eq_applicable(c)) return rc0;
  if (!c->slab || !c->ext_sums) { pa_set_err(c, "pa_bicg_begin is the stepwise form for slabs (pa_slab_set); one GPU: pa_bicgstab"); return PA_E_STATE; }
  PA_HIP(c, hipSetDevice(c->device));
  return c->dtype == PA_F64 ? bicg_slab_begin_t<double>(c, (double*)x, (const double*)rhs, tol, max_it)
                            : bicg_slab_begin_t<float>(c, (float*)x, (const float*)rhs, tol, max_it);
}

int pa_bicg_start(pa_ctx* c) {   // after the all-reduce of sums[1] = r0.r0
  PA_BICG_LIVE(c, "pa_bicg_start")
  if (c->dtype == PA_F64) hipLaunchKernelGGL(k_bicg_logic<double>, dim3(1), dim3(1), 0, c->stream, c->sc, (const double*)pa_sums(c), 10);
  else hipLaunchKernelGGL(k_bicg_logic<float>, dim3(1), dim3(1), 0, c->stream, c->sc, (const double*)pa_sums(c), 10);
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

int pa_bicg_pv(pa_ctx* c) {
  PA_BICG_LIVE(c, "pa_bicg_pv")
  return c->dtype == PA_F64 ? bicg_slab_pv_t<double>(c) : bicg_slab_pv_t<float>(c);
}

int pa_bicg_st(pa_ctx* c) {
  PA_BICG_LIVE(c, "pa_bicg_st")
  return c->dtype == PA_F64 ? bicg_slab_st_t<double>(c) : bicg_slab_st_t<float>(c);
}

int pa_bicg_x(pa_ctx* c) {
  PA_BICG_LIVE(c, "pa_bicg_x")
  return c->dtype == PA_F64 ? bicg_slab_x_t<double>(c) : bicg_slab_x_t<float>(c);
}

int pa_bicg_bc(pa_ctx* c) {
  PA_BICG_LIVE(c, "pa_bicg_bc")
  return c->dtype == PA_F64 ? bicg_slab_bc_t<double>(c) : bicg_slab_bc_t<float>(c);
}

int pa_bicg_finish(pa_ctx* c) {
  PA_BICG_LIVE(c, "pa_bicg_finish")
  if (c->dtype == PA_F64) hipLaunchKernelGGL(k_bicg_logic<double>, dim3(1), dim3(1), 0, c->stream, c->sc, (const double*)pa_sums(c), 3);
  else hipLaunchKernelGGL(k_bicg_logic<float>, dim3(1), dim3(1), 0, c->stream, c->sc, (const double*)pa_sums(c), 3);
  c->cur ^= 1;
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

int pa_bicg_end(pa_ctx* c, pa_report* out) {
  PA_BICG_LIVE(c, "pa_bicg_end")
  int rc = out ? pa_report_read(c, out) : PA_OK;
  c->solver_live = 0;
  if (rc) return rc;
  return (out && out->status) ? PA_E_NONFINITE : PA_OK;
}

}  // extern "C"

// ============================================================================
//  stepwise Jacobi on a slab (SURVEY a15 + 8e): x <- B(x + omega (b - A x) / diag(A)), the CG's stop test
// ============================================================================
// Per sweep: the sweep kernel on the local planes (ghost planes of x through Vec<T>: pa_slab_set's x_ghost_lo / hi) ->
// [exchange of the periodic far planes of the NEW iterate] -> BC fill + shell term, local sum |dx|^2 -> sums[2]
// -> [all-reduce 1] -> stop test (device side) -> [exchange of the new iterate's first / last plane -> x_ghost].
// The iterate ping-pongs between the caller's x and a scratch field; the planes the neighbours need leave through the
// packed send buffers of pa_slab_set (r_send_lo / hi: first / last owned plane; x_pack_*: the periodic far planes).
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_jacobi_rows_to_sum(const SolverScalars* __restrict__ sc, const double* __restrict__ partials,
                                                                  int nblk, const double* __restrict__ partials_shell, int nblk_shell,
                                                                  double* __restrict__ sums) {
  __shared__ double sm[PA_BLOCK / 64];
  if (sc->done) return;
  const double dx2 = pa_reduce_partials(partials, nblk, 2, 1, sm);
  const double sh = nblk_shell > 0 ? pa_reduce_partials(partials_shell, nblk_shell, 1, 0, sm) : 0.0;
  if (threadIdx.x == 0) sums[PA_SUM_DX2] = dx2 + sh;
}

template <typename T>
__global__ void k_jacobi_logic(SolverScalars* sc, const double* __restrict__ sums) {
  if (threadIdx.x != 0 || blockIdx.x != 0 || sc->done) return;
  pa_logic_jacobi<T>(sc, sums[PA_SUM_DX2]);
}

namespace {

template <typename T>
int jacobi_slab_begin_t(pa_ctx* c, T* x, const T* rhs, double tol, int64_t max_it, double omega) {
  const DevGeom& G = c->G;
  for (int q = 0; q < c->nterms; ++q)
    if (c->terms[q].kind != PA_OP_LAPLACIAN) { pa_set_err(c, "pa_jacobi_begin: laplacian terms only"); return PA_E_ARG; }
  const size_t fb = (size_t)G.ncell * sizeof(T);
  int rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_D0], &c->cap[SCR_D0], fb))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_PART], &c->cap[SCR_PART], (size_t)PA_MAX_PARTIALS * 4 * sizeof(double)))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_PART2], &c->cap[SCR_PART2], (size_t)3 * PA_MAX_GRID * sizeof(double)))) return rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_SHELL], &c->cap[SCR_SHELL], 2 * (size_t)pa_shell_elems(c) * sizeof(T)))) return rc;
  if ((rc = init_scalars(c, tol, max_it))) return rc;
  c->bc_static = pa_bc_is_static(c);
  c->bc_fused = pa_bc_fusable(c);
  c->bc_pair = (!c->bc_fused && pa_bc_pairable(c)) ? 1 : 0;
  c->shell_cur = 0;
  c->fold_a_n = c->fold_b_n = c->fold_b_nsh = 0;
  c->cg_pitch = 0;
  c->cg_ps1 = 0;
  // the driver has filled the BCs (it needs the far planes for that) and exchanged the ghost planes of x: only the
  // shell of the start is recorded here (x_old of the first stop test)
  if (c->bc_pair) {
    if ((rc = pa_bc_pair_apply<T>(c, x, nullptr, 3, false, nullptr))) return rc;
  } else {
    pa_shell_launch<T>(c, (const T*)x, (T*)c->scr[SCR_SHELL] + (c->shell_cur ? pa_shell_elems(c) : 0), (double*)c->scr[SCR_PART2], 0);
  }
  c->cg_x = x;
  c->jac_rhs = rhs;
  c->jac_omega = omega;
  c->cur = 0;            // the iterate lives in x (0) or in the scratch field (1)
  c->solver_live = 3;    // (3: the stepwise Jacobi)
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

template <typename T>
int jacobi_slab_sweep_t(pa_ctx* c) {
  const DevGeom& G = c->G;
  const int nblk = pa_grid_blocks(G.ncell);
  DevEq<T> E;
  pa_build_eq<T>(c, c->nterms, c->terms, E);
  T* buf[2] = {(T*)c->cg_x, (T*)c->scr[SCR_D0]};
  const int cur = c->cur;
  Vec<T> xv = slab_vec<T>(c, buf[cur], c->x_glo, c->x_ghi);
  double* part = (double*)c->scr[SCR_PART];
  c->fold_b_n = 0;
  int used = pa_tile3d_jacobi<T>(c, E, xv, (const T*)c->jac_rhs, buf[cur ^ 1], c->jac_omega, part);
  if (used < 0) return used;
  if (used == 0) {
    hipLaunchKernelGGL(k_jacobi<T>, dim3(nblk), dim3(PA_BLOCK), 0, c->stream, G, E, c->sc, xv, (const T*)c->jac_rhs, buf[cur ^ 1],
                       (T)c->jac_omega, part);
    used = nblk;
  }
  c->b_blocks = used;
  // the planes of the NEW iterate the other end of a periodic ring needs for its BC fill
  if (c->x_pack_lo1 || c->x_pack_hi0 || c->x_pack_hi1) {
    const T* xr = (const T*)buf[cur ^ 1];
    hipLaunchKernelGGL(k_pack_planes<T>, dim3(pa_grid_blocks(G.s0)), dim3(PA_BLOCK), 0, c->stream, c->sc, G.s0,
                       xr + 1 * G.s0, (T*)c->x_pack_lo1, xr + (G.n0 - 1) * G.s0, (T*)c->x_pack_hi0,
                       xr + (G.n0 - 2) * G.s0, (T*)c->x_pack_hi1);
  }
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

template <typename T>
int jacobi_slab_bc_t(pa_ctx* c) {
  const DevGeom& G = c->G;
  T* buf[2] = {(T*)c->cg_x, (T*)c->scr[SCR_D0]};
  T* xn = buf[c->cur ^ 1];
  double* part = (double*)c->scr[SCR_PART];
  double* part2 = (double*)c->scr[SCR_PART2];
  int nsh = 0, rc;
  if (!c->bc_static) {
    if (c->bc_fused) {
      if ((rc = pa_bc_shell_fused<T>(c, xn, part2, 1, true, &nsh, false))) return rc;
    } else if (c->bc_pair) {
      if ((rc = pa_bc_pair_apply<T>(c, xn, part2, 1, true, &nsh))) return rc;
    } else {
      if ((rc = pa_bc_apply_faces<T>(c, xn, true))) return rc;
      nsh = pa_shell_blocks(c);
      pa_shell_launch<T>(c, (const T*)xn, (T*)c->scr[SCR_SHELL], part2, 1);
    }
  }
  hipLaunchKernelGGL(k_jacobi_rows_to_sum<T>, dim3(1), dim3(PA_BLOCK), 0, c->stream, c->sc, (const double*)part, c->b_blocks,
                     (const double*)part2, nsh, pa_sums(c));
  // the first / last owned plane of the new iterate, BCs filled: the neighbours' ghost planes of the next sweep
  if (c->r_send_lo || c->r_send_hi)
    hipLaunchKernelGGL(k_pack_planes<T>, dim3(pa_grid_blocks(G.s0)), dim3(PA_BLOCK), 0, c->stream, c->sc, G.s0,
                       (const T*)xn, (T*)c->r_send_lo, (const T*)xn + (G.n0 - 1) * G.s0, (T*)c->r_send_hi, (const T*)nullptr, (T*)nullptr);
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

}  // namespace

extern "C" {

#define PA_JAC_LIVE(c, what)                                                                                          \
  if (!(c) || (c)->solver_live != 3) { if (c) pa_set_err((c), what " without pa_jacobi_begin"); return PA_E_STATE; }   \
  PA_HIP((c), hipSetDevice((c)->device));

int pa_jacobi_begin(pa_ctx* c, void* x, const void* rhs, double tol, int64_t max_it, double omega) {
  if (!c || !c->grid_set || !c->eq_set) { if (c) pa_set_err(c, "pa_jacobi_begin: grid/equation not set"); return PA_E_STATE; }
  if (int rc0 = pa_check_eq_applicable(c)) return rc0;
  if (!c->slab || !c->ext_sums) { pa_set_err(c, "pa_jacobi_begin is the stepwise form for slabs (pa_slab_set); one GPU: pa_jacobi"); return PA_E_STATE; }
  PA_HIP(c, hipSetDevice(c->device));
  return c->dtype == PA_F64 ? jacobi_slab_begin_t<double>(c, (double*)x, (const double*)rhs, tol, max_it, omega)
                            : jacobi_slab_begin_t<float>(c, (float*)x, (const float*)rhs, tol, max_it, omega);
}

int pa_jacobi_sweep(pa_ctx* c) {
  PA_JAC_LIVE(c, "pa_jacobi_sweep")
  return c->dtype == PA_F64 ? jacobi_slab_sweep_t<double>(c) : jacobi_slab_sweep_t<float>(c);
}

int pa_jacobi_bc(pa_ctx* c) {
  PA_JAC_LIVE(c, "pa_jacobi_bc")
  return c->dtype == PA_F64 ? jacobi_slab_bc_t<double>(c) : jacobi_slab_bc_t<float>(c);
}

int pa_jacobi_finish(pa_ctx* c) {   // after the all-reduce of sums[PA_SUM_DX2]
  PA_JAC_LIVE(c, "pa_jacobi_finish")
  if (c->dtype == PA_F64) hipLaunchKernelGGL(k_jacobi_logic<double>, dim3(1), dim3(1), 0, c->stream, c->sc, (const double*)pa_sums(c));
  else hipLaunchKernelGGL(k_jacobi_logic<float>, dim3(1), dim3(1), 0, c->stream, c->sc, (const double*)pa_sums(c));
  c->cur ^= 1;
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

int pa_jacobi_end(pa_ctx* c, pa_report* out) {
  PA_JAC_LIVE(c, "pa_jacobi_end")
  pa_report tmp;
  int rc = pa_report_read(c, out ? out : &tmp);   // synchronises: itr sweeps were executed
  c->solver_live = 0;
  if (rc) return rc;
  // the final iterate lives in the buffer the last EXECUTED sweep wrote (x after an even number of sweeps), the one
  // before it (Field.VARo on request) in the other buffer
  const size_t fb = (size_t)c->G.ncell * (size_t)c->esize;
  void* buf[2] = {c->cg_x, c->scr[SCR_D0]};
  const int fin = (int)(c->h_sc->itr & 1);
  if (c->x_old_out && c->h_sc->itr >= 1)
    PA_HIP(c, hipMemcpyAsync(c->x_old_out, buf[fin ^ 1], fb, hipMemcpyDeviceToDevice, c->stream));
  if (fin) PA_HIP(c, hipMemcpyAsync(c->cg_x, buf[1], fb, hipMemcpyDeviceToDevice, c->stream));
  PA_HIP(c, hipStreamSynchronize(c->stream));
  return (out && out->status) ? PA_E_NONFINITE : PA_OK;
}

}  // extern "C"
