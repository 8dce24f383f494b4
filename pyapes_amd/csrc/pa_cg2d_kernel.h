// pa_cg2d_kernel.h -- k_cg2d: the solver phases on 2-D meshes, marching along the mesh's slow axis (CG phases A / B;
// since the second session of round 3 also the Jacobi sweep, phase 4, and the BiCGSTAB phases 6 (s / t) and 8 (v' = A p'
// from a stored p'): the same strip walk, the outputs / partial sums / prologues of k_cg3d's phases).
//
// k_cg3d treats a 2-D mesh as ONE plane of its tiling: every workgroup handles a 16-row tile once, pays the full
// halo ring (two extra rows, two columns of single-cell loads) for it and never reuses a row -- 4096^2 fp64 CG ran
// at 0.27 ms / iteration = 0.63 of the HBM roofline where 256^3 (the same cell count) reaches 0.87.  Here the slow
// axis of the 2-D mesh (internal axis 1, rows of n2 contiguous cells) is the MARCH axis:
//   * a WAVE owns a strip of 64 x VEC contiguous cells (VEC = 16 B / sizeof(T)) and walks down a chunk of rows; the
//     rows j - 1, j, j + 1 of its strip stay in registers, so every value is loaded once per chunk (the two rows at
//     the chunk ends are the only re-reads);
//   * k +- 1 inside the strip: one DPP move per side (the pa_sf_kernel.h idiom); the cell left / right of the strip
//     is one scalar load per row (lanes 0 and 63 keep it) -- the neighbour wave's own data, an L1 / L2 hit;
//   * no LDS, no barrier; loads run two rows ahead of the arithmetic; the four waves of a workgroup are four
//     adjacent strips; blockIdx -> (chunk, strip group) is XCD-aware like k_cg3d's.
// Arithmetic per node, scalar steps in the prologue (pa_cg_iterate folds them into the kernel that follows), partial
// rows and stop-test terms are those of k_cg3d's CG phases, operation for operation; only the grouping of the partial
// sums differs (per workgroup of this grid).  PITCH: odd row lengths, r / d / d' with a padded row pitch and x cell
// by cell (pa_cg3d_kernel.h, LAY 2).
#pragma once
#include "pa_sf_kernel.h"

#include <algorithm>

// RZ (round 4): axisymmetric meshes.  What differs from xyz is three coefficients of the r axis per r index -- Ap, Am
// and the neumann / symmetry row value (tools.py:86-107, fdc.py:395-417: rows 0 - 2 of pa_coord_set's table) -- and r is
// the MARCH axis here (internal axis 1): a triple per row, wave-uniform, loaded two rows ahead like the row itself.
// Everything else is the xyz kernel, so the arithmetic per node is pa_apply_terms' rz branch operation for operation.
template <typename T, int PHASE, bool PITCH, bool RZ = false>
__global__ void __launch_bounds__(256) k_cg2d(Cg3dArgs<T> A) {
  // PHASE 9 = the Jacobi sweep (4) marching backwards: consecutive sweeps alternate (pa_cg3d_kernel.h); the body tests PH
  constexpr int PH = PHASE == 9 ? 4 : PHASE;
  static_assert(PH == 0 || PH == 1 || PH == 4 || PH == 6 || PH == 8, "CG phases, Jacobi sweep, BiCGSTAB s / t and v phases");
  static_assert(!PITCH || PH != 4, "the Jacobi sweep works on the caller's arrays");
  constexpr int VEC = VecOf<T>::N;
  typedef T V __attribute__((ext_vector_type(VEC)));
  constexpr int TK = 64 * VEC;
  const DevGeom& G = A.G;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int vb = pa_xcd_remap(blockIdx.x, gridDim.x);
  const int groups = A.tiles_k;                       // workgroups side by side along k (4 strips each)
  const int chunk = vb / groups, sg = vb - chunk * groups;
  const int n1 = (int)G.n1, n2 = (int)G.n2;
  const int j0 = (int)((int64_t)chunk * n1 / A.chunks), j1 = (int)((int64_t)(chunk + 1) * n1 / A.chunks);
  const int CJ = j1 - j0;
  constexpr int rev = (PHASE == 1 || PHASE == 9) ? 1 : 0;          // phase B marches backwards (pa_cg3d_kernel.h)
  const int64_t fs1 = PITCH ? A.ps1 : G.s1;           // row stride of r, d, d' (x: G.s1)
  const int64_t k0 = ((int64_t)sg * 4 + wv) * TK;
  const int64_t kg = k0 + (int64_t)lane * VEC;
  const bool kvalid = kg < n2;
  const int64_t kcf = kg < (PITCH ? fs1 : (int64_t)n2) ? kg : 0;     // a lane beyond the row reads column 0 (never used)
  auto wrapk = [&](int64_t v) -> int64_t { v %= n2; return v < 0 ? v + n2 : v; };
  const int64_t ecol = lane == 63 ? wrapk(k0 + TK) : wrapk(k0 - 1);   // lane 63: the cell right of the strip, else left
  auto wrapj = [&](int j) -> int64_t { j %= n1; return j < 0 ? j + n1 : j; };

  // ---- the scalar step folded into this kernel's prologue (k_cg3d, same code: every block reduces the partial
  //      rows the previous kernel left, in the same fixed order -> the same bits in every block) -----------------
  T beta = (T)0, alpha = (T)0;
  if ((PH == 0 || PH == 4) && A.pre_n > 0) {
    __shared__ double pre_sm[16];
    const SolverScalars* si = A.sc;
    const int done_in = si->done;
    const double rr_in = si->rr, beta_in = si->beta, tol_lim = si->tolerance;
    const long long itr_in = si->itr, max_it = si->max_it;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0;
    for (int b = threadIdx.x; b < A.pre_n; b += 256) {
      if (PH == 0) v0 += A.pre_part[2 * (int64_t)b];
      v1 += A.pre_part[2 * (int64_t)b + 1];
    }
    for (int b = threadIdx.x; b < A.pre_nsh; b += 256) v2 += A.pre_shell[b];
    if (done_in) {
      if (blockIdx.x == 0 && threadIdx.x == 0) *A.sc_w = *si;
      return;
    }
    for (int off = 32; off > 0; off >>= 1) {
      v0 += __shfl_down(v0, off, 64);
      v1 += __shfl_down(v1, off, 64);
      v2 += __shfl_down(v2, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      const int w = threadIdx.x >> 6;
      pre_sm[3 * w] = v0;
      pre_sm[3 * w + 1] = v1;
      pre_sm[3 * w + 2] = v2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double rr = 0.0, dx2 = 0.0, sh = 0.0;
      for (int w = 0; w < 4; ++w) { rr += pre_sm[3 * w]; dx2 += pre_sm[3 * w + 1]; sh += pre_sm[3 * w + 2]; }
      const T rr_new = (T)rr;
      const T tolv = (T)sqrt(dx2 + sh);
      const bool bad = isnan(tolv) || isinf(tolv);   // linalg.py:334-336 raises before beta / itr
      const T rr_old = (T)rr_in;
      const double bq = bad ? beta_in : (double)(rr_new / rr_old);
      const long long itr = itr_in + (bad ? 0 : 1);
      const int done = (bad || itr > max_it || !((double)tolv > tol_lim)) ? 1 : 0;
      pre_sm[12] = bq;
      pre_sm[13] = done ? 1.0 : 0.0;
      if (blockIdx.x == 0) {
        SolverScalars* so = A.sc_w;
        *so = *si;
        so->tol = (double)tolv;
        so->done = done;
        if (bad) {
          so->err = 1;
        } else {
          if (PH == 0) {
            so->rr_old = (double)rr_old;
            so->beta = bq;
            so->rr = (double)rr_new;
          }
          so->itr = itr;
        }
        if (PH == 0) A.pre_sums[1] = rr;
        A.pre_sums[2] = dx2 + sh;
      }
    }
    __syncthreads();
    if (pre_sm[13] != 0.0) return;
    beta = (T)pre_sm[12];
  } else if (PH == 1 && A.pre_n > 0) {
    __shared__ double pre_sm[8];
    const int done_in = A.sc->done;
    const double rr_in = A.sc->rr;
    double v0 = 0.0;
    for (int b = threadIdx.x; b < A.pre_n; b += 256) v0 += A.pre_part[b];
    if (done_in) return;
    for (int off = 32; off > 0; off >>= 1) v0 += __shfl_down(v0, off, 64);
    if ((threadIdx.x & 63) == 0) pre_sm[threadIdx.x >> 6] = v0;
    __syncthreads();
    if (threadIdx.x == 0) {
      double v = 0.0;
      for (int w = 0; w < 4; ++w) v += pre_sm[w];
      const T dAd = (T)v;
      const T a = (T)rr_in / dAd;
      const double al = (isnan(a) || isinf(a)) ? 0.0 : (double)a;
      pre_sm[4] = al;
      if (blockIdx.x == 0) {
        A.sc_w->dAd = (double)dAd;
        A.sc_w->alpha = al;
        A.pre_sums[0] = v;
      }
    }
    __syncthreads();
    alpha = (T)pre_sm[4];
  } else if (PH == 8 && A.pre_n > 0) {
    // BiCGSTAB: the step that closes the PREVIOUS iteration (k_cg3d phase 5 / 8 prologue, k_bicg_post stage 3)
    __shared__ double pre_sm[8];
    const SolverScalars* si = A.sc;
    const int done_in = si->done, fe = si->finished_early;
    const double tol_lim = si->tolerance, rho_next = si->rho_next, rho_in = si->rho, alpha_in = si->alpha,
                 omega_in = si->omega;
    const long long itr_in = si->itr, max_it = si->max_it;
    double v0 = 0.0;
    for (int b = threadIdx.x; b < A.pre_n; b += 256) v0 += A.pre_part[b];
    if (done_in) {
      if (blockIdx.x == 0 && threadIdx.x == 0) *A.sc_w = *si;
      return;
    }
    for (int off = 32; off > 0; off >>= 1) v0 += __shfl_down(v0, off, 64);
    if ((threadIdx.x & 63) == 0) pre_sm[threadIdx.x >> 6] = v0;
    __syncthreads();
    if (threadIdx.x == 0) {
      double v = 0.0;
      for (int w = 0; w < 4; ++w) v += pre_sm[w];
      const T tolv = (T)sqrt(v);
      const bool bad = !fe && (isnan(tolv) || isinf(tolv));
      int done = (fe || bad) ? 1 : 0;
      T bq = (T)0;
      if (!fe && !bad) {
        if ((double)tolv <= tol_lim) done = 1;
        if (itr_in >= max_it) done = 1;
        bq = (T)rho_next / (T)rho_in;
        bq = bq * (T)alpha_in;
        bq = bq / (T)omega_in;
      }
      pre_sm[5] = done ? 1.0 : 0.0;
      if (blockIdx.x == 0) {
        SolverScalars* so = A.sc_w;
        *so = *si;
        so->done = done;
        if (!fe) {
          so->tol = (double)tolv;
          if (bad) {
            so->err = 1;
          } else {
            so->beta = (double)bq;
            so->rho = rho_next;
          }
        }
      }
    }
    __syncthreads();
    if (pre_sm[5] != 0.0) return;
  } else if (PH == 6 && A.pre_n > 0) {
    // BiCGSTAB: alpha = rho / (r0 . v) of THIS iteration, iteration count (k_bicg_post stage 0)
    __shared__ double pre_sm[8];
    const int done_in = A.sc->done;
    const double rho_in = A.sc->rho;
    double v0 = 0.0;
    for (int b = threadIdx.x; b < A.pre_n; b += 256) v0 += A.pre_part[b];
    if (done_in) return;
    for (int off = 32; off > 0; off >>= 1) v0 += __shfl_down(v0, off, 64);
    if ((threadIdx.x & 63) == 0) pre_sm[threadIdx.x >> 6] = v0;
    __syncthreads();
    if (threadIdx.x == 0) {
      double v = 0.0;
      for (int w = 0; w < 4; ++w) v += pre_sm[w];
      const T q = (T)rho_in / (T)v;
      const double al = (isnan(q) || isinf(q)) ? 0.0 : (double)q;  // linalg.py:302-305
      pre_sm[4] = al;
      if (blockIdx.x == 0) {
        A.sc_w->itr += 1;   // no other block of this launch reads itr
        A.sc_w->alpha = al;
      }
    }
    __syncthreads();
    alpha = (T)pre_sm[4];
  } else {
    if (A.sc->done) return;
    if (PH == 0) beta = (T)A.sc->beta;
    if (PH == 1 || PH == 6) alpha = (T)A.sc->alpha;
  }

  // ---- per-lane constants of the contiguous axis -------------------------------------------------------------
  unsigned colS = 0, colShell = 0;
  T cPk[VEC], cCk[VEC], cMk[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const int64_t kk = kg + v;
    if (kk < n2 && kk >= G.slo[2] && kk <= G.shi[2]) colS |= 1u << v;
    if (kk == 0 || kk == n2 - 1) colShell |= 1u << v;
    const int rc = pa_row_case(G, 2, kk, G.n2, G.treat);
    cPk[v] = A.lap.inv[2]; cCk[v] = A.lap.m2inv[2]; cMk[v] = A.lap.inv[2];
    if (rc == 1) { cPk[v] = A.lap.c23[2]; cCk[v] = -A.lap.c23[2]; cMk[v] = (T)0; }
    if (rc == 2) { cPk[v] = (T)0; cCk[v] = -A.lap.c23[2]; cMk[v] = A.lap.c23[2]; }
  }
  const T sgn = A.sign, cf = A.coeff;
  const int hasc = A.has_coeff;

  // ---- marching ---------------------------------------------------------------------------------------------
  auto row_of = [&](int q) -> int { return rev ? (j1 - 1 - q) : (j0 + q); };   // may be -1 / n1 (wrapped on load)
  struct Raw { V d, r; T ed, er; };
  auto issue = [&](int jrow, Raw& w) {
    const int64_t jw = wrapj(jrow);
    const T* dp = A.d.p + jw * fs1;
    w.d = *reinterpret_cast<const V*>(dp + kcf);
    w.ed = dp[ecol];
    if (PH == 0 || PH == 6) {
      const T* rp = A.r.p + jw * fs1;
      w.r = *reinterpret_cast<const V*>(rp + kcf);
      w.er = rp[ecol];
    }
  };
  auto finish = [&](const Raw& w, V& e, T& ee) {
    if (PH == 0) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        T bd = beta * w.d[v];
        e[v] = w.r[v] + bd;
      }
      T bd = beta * w.ed;
      ee = w.er + bd;
    } else if (PH == 6) {   // s = r - alpha v (linalg.py:230), on the strip and on its edge cell
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        T av = alpha * w.d[v];
        e[v] = w.r[v] - av;
      }
      T av = alpha * w.ed;
      ee = w.er - av;
    } else {
      e = w.d;
      ee = w.ed;
    }
  };
  V ea, ec, eb;      // behind / current / ahead in march order
  T xa, xe, xb;      // their edge cells (only the current row's is used)
  // RZ: (Ap, Am, row value) of the r axis for the current row, the next one, and the one being loaded
  struct RzRow { T p, m, b; };
  auto rz_row = [&](int jrow) -> RzRow {
    const int64_t jw = wrapj(jrow);
    return RzRow{A.rz[jw], A.rz[A.rz_n + jw], A.rz[2 * A.rz_n + jw]};
  };
  RzRow zc{(T)0, (T)0, (T)0}, zn = zc;
  if constexpr (RZ) { zc = rz_row(row_of(0)); zn = rz_row(row_of(1)); }
  Raw w0, w1, w2;
  issue(row_of(-1), w0);
  issue(row_of(0), w1);
  issue(row_of(1), w2);
  finish(w0, ea, xa);
  finish(w1, ec, xe);
  finish(w2, eb, xb);
  (void)xa;

  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int m = 0; m < CJ; ++m) {
    const int jj = row_of(m);
    V xv, rv;
    if (PH == 1) {
      if (PITCH) {
        rv = __builtin_nontemporal_load(reinterpret_cast<const V*>(A.rw + (int64_t)jj * fs1 + kcf));
        const T* xp = A.x + (int64_t)jj * G.s1;
#pragma unroll
        for (int v = 0; v < VEC; ++v) xv[v] = xp[kg + v < n2 ? kg + v : n2 - 1];
      } else {
        xv = __builtin_nontemporal_load(reinterpret_cast<const V*>(A.x + (int64_t)jj * G.s1 + kcf));
        rv = __builtin_nontemporal_load(reinterpret_cast<const V*>(A.rw + (int64_t)jj * fs1 + kcf));
      }
    }
    if (PH == 4)        // Jacobi: the right-hand side, touched once per sweep (non-temporal, as in k_cg3d)
      xv = __builtin_nontemporal_load(reinterpret_cast<const V*>(A.aux + (int64_t)jj * fs1 + kcf));
    else if (PH == 6 || PH == 8)   // BiCGSTAB: r0 (pitched with the rest)
      xv = __builtin_nontemporal_load(reinterpret_cast<const V*>(A.aux + (int64_t)jj * fs1 + kcf));   // (read once per phase)
    // row m + 2 (wrapped: the rows past the chunk's end are valid memory and their values unused) -- unconditional,
    // a branch between the issue of a load and its use makes the compiler wait for everything outstanding
    Raw w;
    issue(row_of(m + 2), w);
    RzRow zl = zc;
    if constexpr (RZ) zl = rz_row(row_of(m + 2));

    // stencil on row jj (k_cg3d's per-component path; internal axis 0 is inactive on a 2-D mesh)
    const bool jS = jj >= G.slo[1] && jj <= G.shi[1];
    const bool jShell = jj == 0 || jj == n1 - 1;
    T cPj = A.lap.inv[1], cCj = A.lap.m2inv[1], cMj = A.lap.inv[1];
    {
      T cBj = A.lap.c23[1];
      if constexpr (RZ) { cPj = zc.p; cMj = zc.m; cBj = zc.b; }
      const int rc = pa_row_case(G, 1, jj, G.n1, G.treat);
      if (rc == 1) { cPj = cBj; cCj = -cBj; cMj = (T)0; }
      if (rc == 2) { cPj = (T)0; cCj = -cBj; cMj = cBj; }
    }
    const V dn = rev ? ea : eb;   // row jj + 1
    const V up = rev ? eb : ea;   // row jj - 1
    const T left = SfBits<T>::prev(ec[VEC - 1], xe);    // lane 0 keeps its edge cell (left of the strip)
    const T right = SfBits<T>::next(ec[0], xe);         // lane 63 keeps its edge cell (right of the strip)
    V outd, outx;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const T xc = ec[v];
      T ax = (T)0;
      T s = cPj * dn[v];
      T mm = cCj * xc;
      s = s + mm;
      mm = cMj * up[v];
      s = s + mm;
      ax = ax + s;
      const T xpk = (v < VEC - 1) ? ec[v + 1 < VEC ? v + 1 : v] : right;
      const T xmk = (v > 0) ? ec[v > 0 ? v - 1 : 0] : left;
      s = cPk[v] * xpk;
      mm = cCk[v] * xc;
      s = s + mm;
      mm = cMk[v] * xmk;
      s = s + mm;
      ax = ax + s;
      if (hasc) ax = ax * cf;
      ax = ax * sgn;
      const bool inS = jS && (colS >> v & 1);
      if (PH == 0) {
        const T e = inS ? xc : (T)0;
        outd[v] = e;
        T p = e * ax;
        s0 += inS ? (double)p : 0.0;
      } else if (PH == 4) {
        // Jacobi:  x + omega (b - A x) / diag(A)   (k_cg3d phase 4 / k_jacobi, operation for operation)
        T dg = (T)0;
        dg = dg + cCj;
        dg = dg + cCk[v];
        if (hasc) dg = dg * cf;
        dg = dg * sgn;
        T q = xv[v] - ax;
        q = q / dg;
        q = A.p0 * q;
        const T xn = inS ? xc + q : xc;
        const bool offshell = inS && !(jShell || (colShell >> v & 1));
        T df = xn - xc;
        T p2 = df * df;
        s1 += offshell ? (double)p2 : 0.0;
        outd[v] = xn;
      } else if (PH == 6 || PH == 8) {
        // own cells: s (phase 6) everywhere, t = A s / v' = A p' on the interior set (k_cg3d phases 6 / 8)
        const T an = inS ? ax : (T)0;
        outd[v] = xc;
        outx[v] = an;
        const T r0c = xv[v];
        if (PH == 8) {
          T p = r0c * an;
          s0 += inS ? (double)p : 0.0;
        } else {
          T p = xc * xc;
          s0 += (kg + v < n2) ? (double)p : 0.0;   // |s|^2 over every node (tol of linalg.py:233)
          T a = an * xc, b = an * an, cc = r0c * an;
          s1 += inS ? (double)a : 0.0;
          s2 += inS ? (double)b : 0.0;
          s3 += inS ? (double)cc : 0.0;
        }
      } else {
        const T xo = xv[v];
        T ad = alpha * xc;
        T xn = xo + ad;
        T aAd = alpha * ax;
        T rn = rv[v] - aAd;
        xn = inS ? xn : xo;
        rn = inS ? rn : (T)0;
        T p = rn * rn;
        s0 += inS ? (double)p : 0.0;
        const bool offshell = inS && !(jShell || (colShell >> v & 1));
        T df = xn - xo;
        T p2 = df * df;
        s1 += offshell ? (double)p2 : 0.0;
        outd[v] = rn;
        outx[v] = xn;
      }
    }
    if (kvalid) {
      if (PH == 0) {
        __builtin_nontemporal_store(outd, reinterpret_cast<V*>(A.dnew + (int64_t)jj * fs1 + kcf));
      } else if (PH == 4) {
        *reinterpret_cast<V*>(A.out + (int64_t)jj * G.s1 + kcf) = outd;
      } else if (PH == 6) {   // (s leaves only when asked for: k_bicg_x<..., SRV> re-forms it from r and v)
        if (A.out) *reinterpret_cast<V*>(A.out + (int64_t)jj * fs1 + kcf) = outd;
        *reinterpret_cast<V*>(A.out2 + (int64_t)jj * fs1 + kcf) = outx;
      } else if (PH == 8) {
        *reinterpret_cast<V*>(A.out2 + (int64_t)jj * fs1 + kcf) = outx;
      } else {
        __builtin_nontemporal_store(outd, reinterpret_cast<V*>(A.rw_out + (int64_t)jj * fs1 + kcf));
        if (PITCH) {
          T* xp = A.x + (int64_t)jj * G.s1;
#pragma unroll
          for (int v = 0; v < VEC; ++v)
            if (kg + v < n2) xp[kg + v] = outx[v];
        } else {
          __builtin_nontemporal_store(outx, reinterpret_cast<V*>(A.x + (int64_t)jj * G.s1 + kcf));
        }
      }
    }
    ea = ec;
    ec = eb;
    xe = xb;
    finish(w, eb, xb);
    if constexpr (RZ) { zc = zn; zn = zl; }
  }

  if (PH == 0 || PH == 8) {
    double s[1] = {s0};
    pa_block_reduce_store<1>(s, A.partials);
  } else if (PH == 6) {
    double s[4] = {s0, s1, s2, s3};
    pa_block_reduce_store<4>(s, A.partials);
  } else {
    double s[2] = {s0, s1};
    pa_block_reduce_store<2>(s, A.partials);
  }
}

// ---- host side ------------------------------------------------------------------------------------------------
template <typename T, int PHASE, bool PITCH, bool RZ = false>
static int cg2d_blocks_per_cu() {
  static int cached = 0;
  if (!cached) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_cg2d<T, PHASE, PITCH, RZ>, 256, 0) != hipSuccess || n <= 0) n = 4;
    cached = n;
  }
  return cached;
}

// 0: not taken (the caller goes on to k_cg3d's one-plane tiling); else the number of workgroups = partial rows
template <typename T, int PHASE>
static int launch_cg2d(pa_ctx* c, Cg3dArgs<T>& A, bool pitched) {
  constexpr int VEC = VecOf<T>::N;
  const DevGeom& G = c->G;
  const int minrows_env = 0;   // (round 3's measurement knob: chunks of >= this many rows; the rule below is what it found)
  // axisymmetric mesh: this kernel is the ONLY tiled one (k_cg3d's one-plane tiling has no r rows), so it runs from
  // about the size on where the resident solver ends instead of from 1.5 M cells: the alternative is the generic kernels
  const bool rz = c->coord == PA_COORD_RZ;
  // (from 150 k cells on: below that a wave's row walk is latency, and the generic kernels win -- rz 101^2 launch per
  // phase CG 25 us generic against 32 on this kernel; the resident solver takes such meshes anyway)
  const int64_t mincells = rz ? (c->cg2d_mincells < 0 ? -1 : std::min<int64_t>(c->cg2d_mincells, 150000)) : c->cg2d_mincells;   // option "cg2d_mincells"; < 0: never
  if (mincells < 0 || c->ndim != 2 || G.act[0] || A.coeff_f || A.kind != 0 || c->slab) return 0;
  if (G.n1 < 8 || G.n2 < 2 * VEC) return 0;
  if (rz && !c->rz_tab) return 0;
  A.rz = (const T*)c->rz_tab;
  A.rz_n = G.n1;
  // Below ~1.5 M cells the one-plane tiling of k_cg3d stays: a wave here walks its rows one after the other (about
  // a microsecond each), and a small mesh has too few strips x chunks to hide that (measured, fp64 Dirichlet, us per
  // iteration old -> new: 1024^2 23 -> 31, 1536^2 56 -> 49, 2048^2 83 -> 57, 4096^2 265 -> 191, 8192^2 1021 -> 800)
  if (G.n1 * G.n2 < mincells) return 0;
  const int groups = (int)((G.n2 + 4 * 64 * VEC - 1) / (4 * 64 * VEC));
  int bpc;
  if constexpr (PHASE == 4 || PHASE == 9) {   // (the Jacobi sweep is never pitched: the caller's arrays)
    if (pitched) return 0;
    bpc = rz ? cg2d_blocks_per_cu<T, PHASE, false, true>() : cg2d_blocks_per_cu<T, PHASE, false>();
  } else if (rz) {
    bpc = pitched ? cg2d_blocks_per_cu<T, PHASE, true, true>() : cg2d_blocks_per_cu<T, PHASE, false, true>();
  } else {
    bpc = pitched ? cg2d_blocks_per_cu<T, PHASE, true>() : cg2d_blocks_per_cu<T, PHASE, false>();
  }
  const int capacity = cus_of(c) * bpc;
  // rows per chunk: a chunk re-reads two rows, so long chunks where the mesh still fills the chip with them
  // (32 rows: 6 % extra reads), 16 rows otherwise
  int minrows = minrows_env > 0 ? minrows_env : ((int64_t)groups * (G.n1 / 32) >= capacity / 2 ? 32 : 16);
  int chunks = capacity / groups;
  const int maxchunks = (int)std::max<int64_t>(1, G.n1 / minrows);
  if (chunks > maxchunks) chunks = maxchunks;
  if (chunks < 1) chunks = 1;
  const int nblk = groups * chunks;
  if (nblk > PA_MAX_PARTIALS) return 0;
  A.tiles_j = 1;
  A.tiles_k = groups;
  A.chunks = chunks;
  static int dbg = -1;
  if (dbg < 0) dbg = getenv("PYAPES_HIP_DEBUG") ? 4 : 0;
  if (dbg > 0) {
    --dbg;
    fprintf(stderr, "[pyapes_hip] k_cg2d phase %c%s: %d strip groups x %d chunks (~%lld rows each) = %d blocks, %d blocks/CU\n",
            (char)('A' + PHASE), pitched ? " (pitched)" : "", groups, chunks, (long long)(G.n1 / chunks), nblk, bpc);
  }
  if (c->plan_only) return nblk;
  if constexpr (PHASE != 4 && PHASE != 9) {
    if (pitched) {
      if (rz) hipLaunchKernelGGL((k_cg2d<T, PHASE, true, true>), dim3(nblk), dim3(256), 0, c->stream, A);
      else hipLaunchKernelGGL((k_cg2d<T, PHASE, true>), dim3(nblk), dim3(256), 0, c->stream, A);
      return nblk;
    }
  }
  if (rz) hipLaunchKernelGGL((k_cg2d<T, PHASE, false, true>), dim3(nblk), dim3(256), 0, c->stream, A);
  else hipLaunchKernelGGL((k_cg2d<T, PHASE, false>), dim3(nblk), dim3(256), 0, c->stream, A);
  return nblk;
}
