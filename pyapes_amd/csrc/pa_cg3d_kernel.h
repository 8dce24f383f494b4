// pa_cg3d_kernel.h -- the tiled marching kernel k_cg3d and its launch helpers, shared by the two
// translation units that instantiate it (pa_cg3d.hip: CG phases, Euler step, Jacobi sweep;
// pa_cg3d_b.hip: A x and the BiCGSTAB phases) so that they compile in parallel.
#pragma once
// The tiled fast path (2-D / 3-D) of the CG phases, the single-field phases and the BiCGSTAB phases for gfx950 (MI355X).
//
// Equation covered: one Laplacian term (scalar or no coefficient), any BC mix, fp64 / fp32,
// single GPU or slab (ghost planes through Vec<T>).  Anything else returns 0 and the caller
// launches the generic kernels of pa_solver.hip / pa_ops.hip.
//
// Data movement (both phases are HBM-bound; no MFMA):
//   phase A  reads r, d      writes d' = r + beta d          + sum d'.(A d')      3 array passes
//   phase B  reads x, r, d'  writes x += alpha d', r -= alpha A d'  + sums        5 array passes
// A d' is never stored: phase B recomputes it from d' (13 flops) instead of moving 16 B/cell
// through HBM twice, so an iteration moves 8 passes where the algorithmic count is 10.
//
// Tiling.  A workgroup (256 threads = 4 wave64) owns an in-plane tile of TJ = 4*RJ rows by
// TK = 64*VEC contiguous k cells (VEC = 16 B / sizeof(T): every global access is a 16-byte
// lane access, a wave covers 1 KiB of one row) and MARCHES along the slow axis i over a chunk
// of planes.  Per thread the planes i-1, i, i+1 of its own RJ x VEC cells stay in registers;
// the current plane, plus a halo ring (one row above / below, one cell left / right, loaded
// with wrap-around indices = torch.roll semantics), is staged in LDS (double buffered, one
// barrier per plane) for the j+-1 / k+-1 neighbours.  Each value is therefore read from
// HBM/L2 once per chunk; the halo ring and the two extra planes per chunk are the only
// re-reads and are mostly L2 hits because the tiles of one chunk run on one XCD at the same
// time (blockIdx -> (chunk, tile) is XCD-aware: blocks b and b+8 share an XCD's L2).
// Phase B marches the chunk in the opposite direction, so the planes phase A touched last
// (still in the 256 MiB Infinity Cache) are the ones phase B reads first, and vice versa.
//
// The grid is exactly one resident wave of workgroups: chunks = capacity / tiles, so every
// CU carries the same number of identical work items and nothing queues behind a tail.
#include "pa_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

template <typename T> struct VecOf;
template <> struct VecOf<double> { static constexpr int N = 2; typedef double type __attribute__((ext_vector_type(2))); };
template <> struct VecOf<float>  { static constexpr int N = 4; typedef float type __attribute__((ext_vector_type(4))); };

template <typename T>
struct Cg3dArgs {
  DevGeom G;
  LapCoef<T> lap;
  T coeff, sign;
  int has_coeff;
  const T* coeff_f;     // tensor coefficient Gamma(x) of laplacian(Gamma, phi) (fdm.py:166-169) or null
  const SolverScalars* sc;
  Vec<T> r, d;          // phase A: r and the old direction; phase B: d = new direction
  T* dnew;              // phase A output
  T* x;                 // phase B in/out
  T* rw;                // phase B in (residual)
  T* rw_out;            // phase B out: the new residual -- rw itself, or another block when the placement search moves r
                        // (pa_place.hip: a move of r costs no copy this way)
  T* send_lo;           // phase B: copies of r's first / last owned plane (slab) or null
  T* send_hi;
  double* partials;
  int tiles_j, tiles_k, chunks, reverse;
  // single-field phases (2 = A x, 3 = explicit Euler step, 4 = Jacobi sweep): the field is `d`
  const T* aux;         // Jacobi: rhs ; Euler: advection field u (or null) ; BiCGSTAB: r0
  T* out;               // result field
  // BiCGSTAB phases (5: p' = r + beta (p - omega v), v' = A p' ; 6: s = r - alpha v, t = A s)
  Vec<T> v;             // third input field of phase 5 (v) ; phase 6 uses r and d (= v)
  T* out2;              // second result field (v' / t)
  T p0, p1, u;          // Euler: nu, dt, scalar u ; Jacobi: omega
  T hh[3], h2[3], ih[3];  // h, fl(2h), fl(1/h) per axis
  int kind;             // Euler: PA_OP_DIV_*
  int lap_off;          // KIND != 0: the equation is the Div term alone (explicit Div, pure advection)
  GradCoef<T> grd;      // phase 7 (explicit gradient): the row coefficients of k_grad
  int gnd;              // phase 7: mesh dimension (components written: gnd, one field of ncell each)
  // PITCH layout (odd row lengths, below): row / plane strides of the ctx-owned arrays r, d, d'; x keeps G.s0 / G.s1
  int64_t ps0, ps1;
  // axisymmetric mesh (k_cg2d<..., RZ>): the r rows of pa_coord_set's table, n_r entries each
  const T* rz;
  int64_t rz_n;
  int interior_only;    // A x: zero outside the interior set
  int out_all;          // Euler step (k_sf): the caller overwrites every node outside the interior set (its BC
                        // fill covers all 2 * ndim faces), so the step need not preserve phi there
  // BC on load (explicit Euler MARCH on k_sf, pa_sf_kernel.h "BCL"): type of internal face f -- 0 none, 1 dirichlet
  // (bcl_val = g), 2 neumann (bcl_val = the additive constant ((2/3) V) dxf ndir of the fill kernels), 3 symmetry
  int bcl_type[6];
  T bcl_val[6];
  T bcl_c43, bcl_c13;
  // folded scalar step (pre_n > 0; single GPU inside pa_cg_iterate): EVERY block first reduces the partial
  // rows the previous kernel left (same fixed order as k_cg_post_a / k_cg_post_b -> the same bits in
  // every block, no fence, no atomics: the kernel boundary made the rows visible) and runs the scalar
  // logic on registers; block 0 alone stores the state -- phase A into the OTHER scalar slot (sc_w),
  // because blocks of this launch may still be reading `sc`
  const double* pre_part;
  const double* pre_shell;
  int pre_n, pre_nsh;
  SolverScalars* sc_w;
  double* pre_sums;
};

__device__ __forceinline__ int pa_xcd_remap(int b, int nb) {
  // blocks b, b+8, b+16 ... share an XCD: give each XCD a contiguous range of work items
  const int q = nb >> 3, r = nb & 7, xcd = b & 7, w = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + w;
}

__device__ __forceinline__ int64_t pa_wrapmod(int64_t v, int64_t n) {
  v %= n;
  return v < 0 ? v + n : v;
}

// KIND: the Div scheme of the explicit Euler step (PHASE 3) as a compile-time constant -- with all three
// schemes in one body the uniform operands no longer fit the scalar registers and get spilled into
// vector lanes (v_readlane / v_writelane were a third of the VALU instructions of that kernel)
// LAY 1 = NARROW: one cell per lane instead of a 16-byte vector.  For rows whose length is not a multiple of
// the vector width (node-based meshes love 2^k + 1) or operands that are not 16-byte aligned: no
// partial vectors, no alignment demand beyond sizeof(T); everything else is the same code.
// LAY 2 = PITCH (CG phases on one GPU, non-periodic contiguous axis): the same odd rows, but the arrays the ctx
// OWNS -- r and the two direction buffers -- live with a row pitch rounded up to the vector (pa_cg_begin lays them
// out; the pad cells hold 0 and are written as 0), so every access to them is the aligned 16-byte lane access of
// LAY 0.  Only x, the caller's contiguous field, is touched cell by cell: phase A is fully vector, phase B on
// three of its five streams.  A pad cell is never a neighbour anybody uses: the last real cell of a row is a
// boundary node of a non-periodic axis, outside the interior set.  BiCGSTAB (phases 5, 6, 8): EVERY array of these
// phases is the ctx's (r, p, v, r0, s, t) and pitched; only the x / r update (k_bicg_x, pa_solver.hip) touches x.
template <typename T, int RJ, int PHASE, bool CF = false, int KIND = 0, int LAY = 0>
__global__ void __launch_bounds__(256) k_cg3d(Cg3dArgs<T> A) {
  // PHASE 9 = the Jacobi sweep (phase 4) marching its chunks BACKWARDS: consecutive sweeps alternate, so the planes a sweep
  // wrote last -- still in the 256 MiB Infinity Cache -- are the ones the next sweep reads first (what phase B does for
  // phase A).  Everything below tests PH.
  constexpr int PH = PHASE == 9 ? 4 : PHASE;
  constexpr bool NARROW = LAY == 1, PITCH = LAY == 2;
  static_assert(!PITCH || ((PH == 0 || PH == 1 || PH == 5 || PH == 6 || PH == 8) && !CF && KIND == 0),
                "PITCH: the CG / BiCGSTAB phases of a plain Laplacian");
  constexpr int VEC = NARROW ? 1 : VecOf<T>::N;
  typedef T V __attribute__((ext_vector_type(VEC)));
  constexpr int TJ = 4 * RJ, TK = 64 * VEC, TKP = TK + 2 * VEC;
  __shared__ __attribute__((aligned(16))) T tile[2][TJ + 2][TKP];

  (void)sizeof(int[PH >= 0 && PH <= 8 ? 1 : -1]);
  T beta = (T)0, alpha = (T)0;
  T omega = (T)0;
  const DevGeom& G = A.G;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int vb = pa_xcd_remap(blockIdx.x, gridDim.x);
  const int tiles = A.tiles_j * A.tiles_k;
  const int chunk = vb / tiles, tl = vb - chunk * tiles;
  const int tjb = tl / A.tiles_k, tkb = tl - tjb * A.tiles_k;
  const int64_t i0 = (int64_t)chunk * G.n0 / A.chunks, i1 = (int64_t)(chunk + 1) * G.n0 / A.chunks;
  const int CI = (int)(i1 - i0);
  const int64_t j0 = (int64_t)tjb * TJ, k0 = (int64_t)tkb * TK;
  // phase B marches its chunks backwards (header comment); a compile-time fact, so that "ahead" / "behind" resolve to
  // registers instead of 2 * RJ * VEC selects per plane (32 of the 459 VALU instructions of the fp64 phase-A loop)
  // (BiCGSTAB's s / t phase marching backwards as well -- between the forward v phase and the x / r update -- measured in
  // round 4, interleaved on one box: 256^3 0.3984 / 0.3984 ms, 512^3 3.069 / 3.059, 2.973 / 2.969: nothing, not kept)
  constexpr int rev = (PHASE == 1 || PHASE == 9) ? 1 : 0;

  // ---- per-thread geometry: RJ rows x VEC columns --------------------------------
  const int64_t kg = k0 + (int64_t)lane * VEC;            // global k of element 0 (may be >= n2)
  const int64_t kc = pa_wrapmod(kg, G.n2);                // wrapped column used for loads
  const bool kvalid = kg < G.n2;
  // strides / column of the arrays staged through LDS (r, d, v): pitched rows in the PITCH layout, where a lane
  // beyond the padded row reads column 0 (aligned; its values are never used)
  const int64_t fs0 = PITCH ? A.ps0 : G.s0, fs1 = PITCH ? A.ps1 : G.s1;
  const int64_t kcf = PITCH ? (kg < fs1 ? kg : 0) : kc;
  int64_t jrow[RJ];
  unsigned rowS = 0, rowShell = 0, rowValid = 0, rowLo = 0, rowHi = 0;
#pragma unroll
  for (int jj = 0; jj < RJ; ++jj) {
    const int64_t jg = j0 + wv * RJ + jj;
    jrow[jj] = pa_wrapmod(jg, G.n1);
    const bool valid = jg < G.n1;
    if (valid) rowValid |= 1u << jj;
    if (valid && jg >= G.slo[1] && jg <= G.shi[1]) rowS |= 1u << jj;
    if (jg == 0 || jg == G.n1 - 1) rowShell |= 1u << jj;
    const int rc = pa_row_case(G, 1, jg, G.n1, G.treat);
    if (rc == 1) rowLo |= 1u << jj;
    if (rc == 2) rowHi |= 1u << jj;
  }
  unsigned rowPLo = 0, rowPHi = 0, colPLo = 0, colPHi = 0;  // periodic rows of the central Div (fdc.py:596-602)
  if (PH == 3 || PH == 7 || KIND != 0) {
#pragma unroll
    for (int jj = 0; jj < RJ; ++jj) {
      const int64_t jg = j0 + wv * RJ + jj;
      if (G.bct[2] == 4 && jg == 1) rowPLo |= 1u << jj;
      if (G.bct[3] == 4 && jg == G.n1 - 2) rowPHi |= 1u << jj;
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      if (G.bct[4] == 4 && kg + v == 1) colPLo |= 1u << v;
      if (G.bct[5] == 4 && kg + v == G.n2 - 2) colPHi |= 1u << v;
    }
  }
  unsigned colS = 0, colShell = 0, colLo = 0, colHi = 0;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const int64_t kk = kg + v;
    if (kvalid && kk >= G.slo[2] && kk <= G.shi[2]) colS |= 1u << v;
    if (kk == 0 || kk == G.n2 - 1) colShell |= 1u << v;
    const int rc = pa_row_case(G, 2, kk, G.n2, G.treat);
    if (rc == 1) colLo |= 1u << v;
    if (rc == 2) colHi |= 1u << v;
  }
#ifdef PA_PLAIN_EXPERIMENT   // measurement only: what the phases cost without their per-row / per-column case logic
  rowLo = rowHi = colLo = colHi = 0;
  rowS = rowValid = (1u << RJ) - 1;
  colS = (1u << VEC) - 1;
#endif
  // halo duty of this wave: wave 0 -> row above the tile, wave 3 -> row below (vector loads);
  // wave 1, lanes < 2*TJ -> the single cells left / right of each tile row (scalar loads)
  const int64_t hrow = (wv == 0) ? pa_wrapmod(j0 - 1, G.n1) : pa_wrapmod(j0 + TJ, G.n1);
  const bool hvec = (wv == 0 || wv == 3);
  const bool hsc = (wv == 1 && lane < 2 * TJ);
  const int hs_row = lane >> 1, hs_side = lane & 1;
  const int64_t hs_off = pa_wrapmod(j0 + hs_row, G.n1) * fs1 +
                         (hs_side ? pa_wrapmod(k0 + TK, G.n2) : pa_wrapmod(k0 - 1, G.n2));

  auto plane_of = [&](int q) -> int64_t { return rev ? (i1 - 1 - q) : (i0 + q); };
  auto pptr = [&](const Vec<T>& v, int64_t ii) -> const T* {
    return ii < 0 ? v.glo : (ii >= G.n0 ? v.ghi : v.p + ii * fs0);
  };

  // Raw loads of one plane (own cells + this wave's share of the halo ring).  They are only
  // ISSUED here; the arithmetic that consumes them (finish_*) is placed after the stencil of the
  // current plane, so the s_waitcnt lands there and the loads fly during the stencil.
  struct Raw {
    V d[RJ];
    V r[RJ];   // phases A, 5, 6
    V q[RJ];   // phase 5 (v)
    V hd, hr, hq;  // halo row (waves 0 and 3)
    T sd, sr, sq;  // halo cell (wave 1)
  };
  constexpr bool HAS_R = (PH == 0 || PH == 5 || PH == 6);
  constexpr bool HAS_Q = (PH == 5);
  auto issue_at = [&](const T* dp, const T* rp, const T* qp, Raw& w, bool with_halo) {
#pragma unroll
    for (int jj = 0; jj < RJ; ++jj) {
      const int64_t o = jrow[jj] * fs1 + kcf;
      w.d[jj] = *reinterpret_cast<const V*>(dp + o);
      if (HAS_R) w.r[jj] = *reinterpret_cast<const V*>(rp + o);
      if (HAS_Q) w.q[jj] = *reinterpret_cast<const V*>(qp + o);
    }
    if (with_halo) {
      if (hvec) {
        const int64_t o = hrow * fs1 + kcf;
        w.hd = *reinterpret_cast<const V*>(dp + o);
        if (HAS_R) w.hr = *reinterpret_cast<const V*>(rp + o);
        if (HAS_Q) w.hq = *reinterpret_cast<const V*>(qp + o);
      }
      if (hsc) {
        w.sd = dp[hs_off];
        if (HAS_R) w.sr = rp[hs_off];
        if (HAS_Q) w.sq = qp[hs_off];
      }
    }
  };
  auto issue = [&](int64_t ii, Raw& w, bool with_halo) {
    const T* dp = pptr(A.d, ii);
    issue_at(dp, HAS_R ? pptr(A.r, ii) : dp, HAS_Q ? pptr(A.v, ii) : dp, w, with_halo);
  };
  // the staged field from the raw loads: phase A r + beta d ; phase 5 r + beta (p - omega v)
  // (linalg.py:217) ; phase 6 r - alpha v (linalg.py:230) ; else the field itself
  auto combine = [&](T rr_, T dd_, T qq_) -> T {
    if (PH == 0) {
      T bd = beta * dd_;
      return rr_ + bd;
    } else if (PH == 5) {
      T t = omega * qq_;
      t = dd_ - t;
      t = beta * t;
      return rr_ + t;
    } else if (PH == 6) {
      T av = alpha * dd_;
      return rr_ - av;
    }
    return dd_;
  };
  auto finish_own = [&](const Raw& w, V (&e)[RJ]) {
#pragma unroll
    for (int jj = 0; jj < RJ; ++jj) {
#pragma unroll
      for (int v = 0; v < VEC; ++v)
        e[jj][v] = combine(HAS_R ? w.r[jj][v] : (T)0, w.d[jj][v], HAS_Q ? w.q[jj][v] : (T)0);
    }
  };
  auto finish_halo = [&](const Raw& w, V& hv, T& hs) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) hv[v] = combine(HAS_R ? w.hr[v] : (T)0, w.hd[v], HAS_Q ? w.hq[v] : (T)0);
    hs = combine(HAS_R ? w.sr : (T)0, w.sd, HAS_Q ? w.sq : (T)0);
  };
  auto stage = [&](int buf, const V (&e)[RJ], const V& hv, const T& hs) {
#pragma unroll
    for (int jj = 0; jj < RJ; ++jj)
      *reinterpret_cast<V*>(&tile[buf][wv * RJ + jj + 1][VEC + lane * VEC]) = e[jj];
    if (hvec) *reinterpret_cast<V*>(&tile[buf][wv == 0 ? 0 : TJ + 1][VEC + lane * VEC]) = hv;
    if (hsc) tile[buf][hs_row + 1][hs_side ? VEC + TK : VEC - 1] = hs;
  };

  V ea[RJ], ec[RJ], eb[RJ];  // behind / current / ahead in march order
  V hv;
  T hs = (T)0;
  Raw w;
#pragma unroll
  for (int v = 0; v < VEC; ++v) { hv[v] = (T)0; w.hd[v] = (T)0; w.hr[v] = (T)0; w.hq[v] = (T)0; }
  w.sd = (T)0; w.sr = (T)0; w.sq = (T)0;

  // ---- prologue -----------------------------------------------------------------------
  // All loads a workgroup needs before its first stencil are issued back to back -- the three planes
  // behind / at / ahead of the chunk start, then (below) the flag, the scalars and the partial rows of
  // the folded scalar step -- and waited for once instead of one `finish` at a time (five dependent
  // round trips).  Worth little, as it turned out: 64^3 fp64 CG 20.9 -> 20.1 us per iteration, 128^3
  // 39.7 -> 39.3 -- the trips hit in L2; what is left of a ~10 us kernel on such meshes is its dispatch.
  const bool act0 = G.act[0] != 0;   // 2-D meshes occupy internal axes 1,2: a single plane, no i-neighbours
  Raw wa = w, wc = w;                // planes -1 and 0; `w` takes plane +1 and stays the loop's buffer
  if (act0) issue(plane_of(-1), wa, false);
  issue(plane_of(0), wc, true);
  if (act0) issue(plane_of(1), w, CI > 1);

  if ((PH == 0 || PH == 4) && A.pre_n > 0) {
    // the scalar step that closes the PREVIOUS iteration (CG: linalg.py:128-141, 321-338; the Jacobi
    // sweep has the stop test and the iteration count only).  Every load of
    // the prologue is issued before the first wait -- one memory round trip (~1-2 us right after a
    // kernel boundary), not one per reduction.  Summation order = pa_reduce_partials, column by column.
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
      // pa_logic_b on registers: a local SolverScalars puts the kernel on scratch memory, and a kernel
      // with a private segment costs ~10 us more to dispatch (measured)
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
    // alpha = r.r / d.Ad of THIS iteration (linalg.py:118-120); loads first, as above
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
  } else if ((PH == 5 || PH == 8) && A.pre_n > 0) {
    // BiCGSTAB: the step that closes the PREVIOUS iteration (k_bicg_post stage 3; linalg.py:212-214,
    // 236-262): early exit, stop test 2, next beta, rho <- rho_next.  Next state -> the other slot.
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
      pre_sm[4] = (double)bq;
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
    beta = (T)pre_sm[4];
    omega = (T)omega_in;
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
    if (PH != 2 && PH != 3 && PH != 7 && A.sc->done) return;
    if (PH == 0) beta = (T)A.sc->beta;
    if (PH == 1) alpha = (T)A.sc->alpha;
    if (PH == 5) { beta = (T)A.sc->beta; omega = (T)A.sc->omega; }
    if (PH == 6) alpha = (T)A.sc->alpha;
  }

  if (act0) {
    finish_own(wa, ea);
  } else {
#pragma unroll
    for (int jj = 0; jj < RJ; ++jj)
#pragma unroll
      for (int v = 0; v < VEC; ++v) ea[jj][v] = (T)0;
  }
  finish_own(wc, ec);
  finish_halo(wc, hv, hs);
  stage(0, ec, hv, hs);
  if (act0) {
    finish_own(w, eb);
    finish_halo(w, hv, hs);
  } else {
#pragma unroll
    for (int jj = 0; jj < RJ; ++jj)
#pragma unroll
      for (int v = 0; v < VEC; ++v) eb[jj][v] = (T)0;
  }
  __syncthreads();

  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  const T sgn = A.sign, cf = A.coeff;
  const int hasc = A.has_coeff;
  const T cfe = hasc ? cf : (T)1;   // (x * 1 is x, bit for bit: one multiply instead of a select per cell in the solver phases)
  // the stencil works on whole V rows (packed fp32 multiplies / adds; the explicit Euler step moves only
  // 8 B / cell and is VALU-bound when written per component): per-component k-axis coefficients
  // CG / BiCGSTAB phases of a pure Laplacian: per component (A/B: -9 % on fp32 CG as V rows)
  constexpr bool VROW = (PH == 2 || PH == 3 || PH == 4 || PH == 7 || KIND != 0);
  V cPkV, cCkV, cMkV;
  if (VROW) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      T p = A.lap.inv[2], c0 = A.lap.m2inv[2], mq = A.lap.inv[2];
      if (colLo >> v & 1) { p = A.lap.c23[2]; c0 = -A.lap.c23[2]; mq = (T)0; }
      if (colHi >> v & 1) { p = (T)0; c0 = -A.lap.c23[2]; mq = A.lap.c23[2]; }
      cPkV[v] = p; cCkV[v] = c0; cMkV[v] = mq;
    }
  }
  V gPkV, gCkV, gMkV;   // phase 7: k-axis rows of the gradient (k_grad, pa_ops.hip)
  if (PH == 7) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      T p = A.grd.g[2], c0 = (T)0, mq = A.grd.mg[2];
      if (colLo >> v & 1) { p = A.grd.lo_p[2]; c0 = A.grd.lo_c[2]; mq = (T)0; }
      if (colHi >> v & 1) { p = (T)0; c0 = A.grd.hi_c[2]; mq = A.grd.hi_m[2]; }
      if (colPLo >> v & 1) mq = (T)0;
      if (colPHi >> v & 1) p = (T)0;
      gPkV[v] = p; gCkV[v] = c0; gMkV[v] = mq;
    }
  }

  // Plane pointers of the loads issued inside the loop (plane m + 2 of the march).  Every plane but the one BEYOND the
  // chunk's last lies inside the array: a running pointer, one scalar add per plane; the plane beyond -- the only one
  // that can be a ghost / wrap-around plane -- is resolved once, here.  (Resolved per plane through pptr() the three
  // Vec<T> of the kernel argument were re-read from spilled scalar registers in every iteration: 64 v_readlane of the
  // 459 VALU instructions of the fp64 phase-A loop.)
  const int64_t pstep = rev ? -fs0 : fs0;
  const T* d_run = A.d.p + plane_of(1) * fs0;
  const T* r_run = HAS_R ? A.r.p + plane_of(1) * fs0 : d_run;
  const T* q_run = HAS_Q ? A.v.p + plane_of(1) * fs0 : d_run;
  const T* const d_end = pptr(A.d, plane_of(CI));
  const T* const r_end = HAS_R ? pptr(A.r, plane_of(CI)) : d_end;
  const T* const q_end = HAS_Q ? pptr(A.v, plane_of(CI)) : d_end;

  for (int m = 0; m < CI; ++m) {
    const int buf = m & 1;
    const int64_t ii = plane_of(m);
    const bool more = m + 1 < CI;
    // plane m+1 into the other LDS buffer (its last readers passed the barrier of step m-1)
    if (more) stage(buf ^ 1, eb, hv, hs);
    // phase B: the thread's x and r of THIS plane, issued first so that their wait (at the update,
    // below the stencil) does not have to cover the younger loads of plane m+2
    V xv[RJ], rv[RJ];
    if (PH == 1) {
#pragma unroll
      for (int jj = 0; jj < RJ; ++jj) {
        const int64_t o = ii * G.s0 + jrow[jj] * G.s1 + kc;
        // x and r are touched exactly once per iteration: non-temporal, so that they do not evict the
        // halo rows / planes of d' other workgroups are about to re-read from L2 (measured -3 % on
        // the iteration at 512^3; non-temporal loads of d / r themselves cost +6 % in phase A)
        if (PITCH) {    // r: pitched, aligned vector; x: the caller's contiguous rows, cell by cell
          rv[jj] = __builtin_nontemporal_load(reinterpret_cast<const V*>(A.rw + ii * fs0 + jrow[jj] * fs1 + kcf));
          const T* xp = A.x + ii * G.s0 + jrow[jj] * G.s1;
#pragma unroll
          for (int v = 0; v < VEC; ++v) xv[jj][v] = xp[kg + v < G.n2 ? kg + v : G.n2 - 1];
        } else if (NARROW) {   // 8-byte / 4-byte lanes: the streaming hint costs 8-17 % here (interleaved A/B)
          xv[jj] = *reinterpret_cast<const V*>(A.x + o);
          rv[jj] = *reinterpret_cast<const V*>(A.rw + o);
        } else {
          xv[jj] = __builtin_nontemporal_load(reinterpret_cast<const V*>(A.x + o));
          rv[jj] = __builtin_nontemporal_load(reinterpret_cast<const V*>(A.rw + o));
        }
      }
    }
    V cv[RJ];
    if (CF) {
#pragma unroll
      for (int jj = 0; jj < RJ; ++jj)
        cv[jj] = *reinterpret_cast<const V*>(A.coeff_f + ii * G.s0 + jrow[jj] * G.s1 + kc);
    }
    if (PH == 4 || PH == 5 || PH == 6 || PH == 8 || ((PH == 3 || (PH == 2 && KIND != 0)) && A.aux)) {  // rhs / u / r0 of this plane
#pragma unroll
      for (int jj = 0; jj < RJ; ++jj) {
        // BiCGSTAB's r0 is read once per phase and never written: non-temporal (round 4, three interleaved pairs: 256^3 fp64
        // 0.3854 -> 0.3765 ms / iteration, 512^3 3.098 -> 3.055; 8-byte lanes of the NARROW kernels: not, as everywhere)
        if ((PH == 5 || PH == 6 || PH == 8) && !NARROW) {
          xv[jj] = PITCH ? __builtin_nontemporal_load(reinterpret_cast<const V*>(A.aux + ii * fs0 + jrow[jj] * fs1 + kcf))
                         : __builtin_nontemporal_load(reinterpret_cast<const V*>(A.aux + ii * G.s0 + jrow[jj] * G.s1 + kc));
          continue;
        }
        xv[jj] = PITCH ? *reinterpret_cast<const V*>(A.aux + ii * fs0 + jrow[jj] * fs1 + kcf)   // (BiCGSTAB: r0, pitched)
                 // (the Jacobi right-hand side is touched once per sweep: non-temporal, so that it does not push the iterate
                 // the next sweep starts on out of the Infinity Cache -- 256^3 fp64 0.0845 -> 0.0810 ms / sweep, 512^3 -1 %)
                 : (PH == 4 && !NARROW) ? __builtin_nontemporal_load(reinterpret_cast<const V*>(A.aux + ii * G.s0 + jrow[jj] * G.s1 + kc))
                       : *reinterpret_cast<const V*>(A.aux + ii * G.s0 + jrow[jj] * G.s1 + kc);
      }
    }
    // loads of plane m+2 (own cells + halo): in flight during the stencil below
    d_run += pstep;
    if (HAS_R) r_run += pstep;
    if (HAS_Q) q_run += pstep;
    if (more) {
      const bool inside = m + 2 < CI;
      issue_at(inside ? d_run : d_end, HAS_R ? (inside ? r_run : r_end) : d_run, HAS_Q ? (inside ? q_run : q_end) : d_run, w, inside);
    }

    // ---- stencil on plane ii, every cell of the thread (masks are applied afterwards) --------
    const int64_t gi = ii + G.off0;
    const bool iS = gi >= G.slo[0] && gi <= G.shi[0];
    const bool iShell = act0 && (gi == 0 || gi == G.g0 - 1);
    T cPi = A.lap.inv[0], cCi = A.lap.m2inv[0], cMi = A.lap.inv[0];
    {
      const int rc = pa_row_case(G, 0, gi, G.g0, G.treat);
      if (rc == 1) { cPi = A.lap.c23[0]; cCi = -A.lap.c23[0]; cMi = (T)0; }
      if (rc == 2) { cPi = (T)0; cCi = -A.lap.c23[0]; cMi = A.lap.c23[0]; }
      // a 2-D mesh has no i axis: its planes behind / ahead are +0 (prologue), and with zero coefficients the i term of
      // the solver phases is +0 exactly -- (+0)(+0) + (+0) x + (+0)(+0) -- as the select it replaces gave
      if (!act0) { cPi = (T)0; cCi = (T)0; cMi = (T)0; }
    }
    const bool iPLo = (PH == 3 || PH == 7 || KIND != 0) && G.bct[0] == 4 && gi == 1;
    const bool iPHi = (PH == 3 || PH == 7 || KIND != 0) && G.bct[1] == 4 && gi == G.g0 - 2;
    T gP0 = A.grd.g[0], gC0 = (T)0, gM0 = A.grd.mg[0];
    if (PH == 7) {
      const int rc = pa_row_case(G, 0, gi, G.g0, G.treat);
      if (rc == 1) { gP0 = A.grd.lo_p[0]; gC0 = A.grd.lo_c[0]; gM0 = (T)0; }
      if (rc == 2) { gP0 = (T)0; gC0 = A.grd.hi_c[0]; gM0 = A.grd.hi_m[0]; }
      if (iPLo) gM0 = (T)0;
      if (iPHi) gP0 = (T)0;
    }
    V res[RJ];
    V g0r[RJ], g1r[RJ], g2r[RJ];   // phase 7: the three components of this plane's rows
#pragma unroll
    for (int jj = 0; jj < RJ; ++jj) {
      const int R = wv * RJ + jj + 1;
      T cPj = A.lap.inv[1], cCj = A.lap.m2inv[1], cMj = A.lap.inv[1];
      if (rowLo >> jj & 1) { cPj = A.lap.c23[1]; cCj = -A.lap.c23[1]; cMj = (T)0; }
      if (rowHi >> jj & 1) { cPj = (T)0; cCj = -A.lap.c23[1]; cMj = A.lap.c23[1]; }
      // j-1 / j+1: registers inside the thread's row block, LDS across it
      V up, dn;
      if (jj > 0) up = ec[jj - 1]; else up = *reinterpret_cast<const V*>(&tile[buf][R - 1][VEC + lane * VEC]);
      if (jj < RJ - 1) dn = ec[jj + 1]; else dn = *reinterpret_cast<const V*>(&tile[buf][R + 1][VEC + lane * VEC]);
      const T left = tile[buf][R][VEC + lane * VEC - 1];
      const T right = tile[buf][R][VEC + lane * VEC + VEC];
      if constexpr (VROW) {
        // A x of this row, one V at a time: the same operations in the same order as a per-component loop
        const V xc = ec[jj];
        const V xpi = rev ? ea[jj] : eb[jj];
        const V xmi = rev ? eb[jj] : ea[jj];
        V xpk, xmk;
  #pragma unroll
        for (int v = 0; v < VEC; ++v) {
          xpk[v] = (v < VEC - 1) ? xc[v + 1 < VEC ? v + 1 : v] : right;
          xmk[v] = (v > 0) ? xc[v > 0 ? v - 1 : 0] : left;
        }
        if constexpr (PH == 7) {
          // grad: y[a] = cP xp + cC xc + cM xm with the rows of k_grad (fdc.py:80-87, 543-609), same order
          T pj = A.grd.g[1], cj = (T)0, mj = A.grd.mg[1];
          if (rowLo >> jj & 1) { pj = A.grd.lo_p[1]; cj = A.grd.lo_c[1]; mj = (T)0; }
          if (rowHi >> jj & 1) { pj = (T)0; cj = A.grd.hi_c[1]; mj = A.grd.hi_m[1]; }
          if (rowPLo >> jj & 1) mj = (T)0;
          if (rowPHi >> jj & 1) pj = (T)0;
          V s = gP0 * xpi;
          V mm = gC0 * xc;
          s = s + mm;
          mm = gM0 * xmi;
          g0r[jj] = s + mm;
          s = pj * dn;
          mm = cj * xc;
          s = s + mm;
          mm = mj * up;
          g1r[jj] = s + mm;
          s = gPkV * xpk;
          mm = gCkV * xc;
          s = s + mm;
          mm = gMkV * xmk;
          g2r[jj] = s + mm;
          continue;
        }
        V axv;
        {
          V s = cPi * xpi;
          V mm = cCi * xc;
          s = s + mm;
          mm = cMi * xmi;
          s = s + mm;
          if (act0) axv = s; else axv = (V)(T)0;
          s = cPj * dn;
          mm = cCj * xc;
          s = s + mm;
          mm = cMj * up;
          s = s + mm;
          axv = axv + s;
          s = cPkV * xpk;
          mm = cCkV * xc;
          s = s + mm;
          mm = cMkV * xmk;
          s = s + mm;
          axv = axv + s;
          if (hasc) {
            if (CF) axv = axv * cv[jj]; else axv = axv * cf;
          }
          axv = axv * sgn;
        }
        // Div(u phi) of this row with the scheme KIND (fdc.py:708-772; 4 = upwind as tests/test_fdm.py:239
        // states it): shared by the explicit Euler step and the Laplacian + Div operators
        auto div_row = [&](const V& uc) -> V {
          const V xp3[3] = {xpi, dn, xpk}, xm3[3] = {xmi, up, xmk};
          V adv = (V)(T)0;
          if (KIND == 4) {  // upwind as the reference's test states it
            V upl, umi;
  #pragma unroll
            for (int v = 0; v < VEC; ++v) {
              upl[v] = uc[v] > (T)0 ? uc[v] : (T)0;
              umi[v] = uc[v] < (T)0 ? uc[v] : (T)0;
            }
  #pragma unroll
            for (int a = 0; a < 3; ++a) {
              if (a == 0 && !act0) continue;
              V bwd = xc - xm3[a];
              V fwd = xp3[a] - xc;
              V t = upl * bwd;
              V m2 = umi * fwd;
              t = t + m2;
              t = t * A.ih[a];
              adv = adv + t;
            }
          } else if (KIND == 3) {  // literal reference upwind
            V cP, cC, cM;
  #pragma unroll
            for (int v = 0; v < VEC; ++v) {
              cP[v] = (T)2 * (uc[v] < (T)0 ? uc[v] : (T)0);
              cC[v] = (T)0 * ((T)2 * uc[v]);
              cM[v] = (T)2 * (uc[v] > (T)0 ? uc[v] : (T)0);
            }
  #pragma unroll
            for (int a = 0; a < 3; ++a) {
              if (a == 0 && !act0) continue;
              V t = cP * xp3[a];
              V m2 = cC * xc;
              t = t + m2;
              m2 = cM * xm3[a];
              t = t + m2;
              adv = adv + t;
            }
          } else {  // central, scalar u
  #pragma unroll
            for (int a = 0; a < 3; ++a) {
              if (a == 0 && !act0) continue;
              V cP = uc, cC = (T)0 * uc, cM = -uc;
  #pragma unroll
              for (int v = 0; v < VEC; ++v) {
                const bool lo = a == 0 ? iPLo : (a == 1 ? (bool)(rowPLo >> jj & 1) : (bool)(colPLo >> v & 1));
                const bool hi = a == 0 ? iPHi : (a == 1 ? (bool)(rowPHi >> jj & 1) : (bool)(colPHi >> v & 1));
                if (lo) cM[v] = (T)0;
                if (hi) cP[v] = (T)0;
              }
              cP = cP / A.h2[a];
              cC = cC / A.h2[a];
              cM = cM / A.h2[a];
              V t = cP * xp3[a];
              V m2 = cC * xc;
              t = t + m2;
              m2 = cM * xm3[a];
              t = t + m2;
              adv = adv + t;
            }
          }
          return adv;
        };
        if constexpr (PH == 3) {
          const V ax = axv;
          V uc;
          if (A.aux) uc = xv[jj]; else uc = (V)A.u;
          const V adv = div_row(uc);
          V q = A.p0 * ax;
          q = q - adv;
          q = A.p1 * q;
          res[jj] = xc + q;
          continue;
        }
        if constexpr (KIND != 0 && PH != 3) {
          // sum_k sign_k Aop_k (ops.py:122-154) of {Laplacian, Div(scalar u)}: p0 = sign of the Div term,
          // p1 != 0: the Div term comes first in the equation
          V dv = div_row((PH == 2 && A.aux) ? xv[jj] : (V)A.u);   // A x: the speed may be a field (upwind)
          dv = dv * A.p0;
          if (A.lap_off) axv = (V)(T)0;   // Div alone: 0 + sign * Div, as the generic kernel's running sum
          if (A.p1 != (T)0) axv = dv + axv; else axv = axv + dv;
        }
  #pragma unroll
        for (int v = 0; v < VEC; ++v) {
          T ax = axv[v];
          T cCk = cCkV[v];
          if (PH == 4) {
            // Jacobi:  x + omega (b - A x) / diag(A)   (k_jacobi, pa_solver.hip)
            T dg = act0 ? cCi : (T)0;
            dg = dg + cCj;
            dg = dg + cCk;
            if (hasc) dg = dg * (CF ? cv[jj][v] : cf);
            dg = dg * sgn;
            T q = xv[jj][v] - ax;
            q = q / dg;
            q = A.p0 * q;
            ax = xc[v] + q;
          }
          res[jj][v] = ax;
        }
      } else {
  #pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const T xc = ec[jj][v];
          const T xpi = rev ? ea[jj][v] : eb[jj][v];
          const T xmi = rev ? eb[jj][v] : ea[jj][v];
          T s = cPi * xpi;
          T mm = cCi * xc;
          s = s + mm;
          mm = cMi * xmi;
          s = s + mm;
          T ax = s;
          s = cPj * dn[v];
          mm = cCj * xc;
          s = s + mm;
          mm = cMj * up[v];
          s = s + mm;
          ax = ax + s;
          T cPk = A.lap.inv[2], cCk = A.lap.m2inv[2], cMk = A.lap.inv[2];
          if (colLo >> v & 1) { cPk = A.lap.c23[2]; cCk = -A.lap.c23[2]; cMk = (T)0; }
          if (colHi >> v & 1) { cPk = (T)0; cCk = -A.lap.c23[2]; cMk = A.lap.c23[2]; }
          const T xpk = (v < VEC - 1) ? ec[jj][v + 1 < VEC ? v + 1 : v] : right;
          const T xmk = (v > 0) ? ec[jj][v > 0 ? v - 1 : 0] : left;
          s = cPk * xpk;
          mm = cCk * xc;
          s = s + mm;
          mm = cMk * xmk;
          s = s + mm;
          ax = ax + s;
          if (CF) ax = ax * cv[jj][v]; else ax = ax * cfe;
          ax = ax * sgn;
          if (PH == 4) {
            // Jacobi:  x + omega (b - A x) / diag(A)   (k_jacobi, pa_solver.hip)
            T dg = act0 ? cCi : (T)0;
            dg = dg + cCj;
            dg = dg + cCk;
            if (hasc) dg = dg * (CF ? cv[jj][v] : cf);
            dg = dg * sgn;
            T q = xv[jj][v] - ax;
            q = q / dg;
            q = A.p0 * q;
            ax = xc + q;
          }
          res[jj][v] = ax;
        }
      }
    }

    // ---- outputs of plane ii --------------------------------------------------------------------
#pragma unroll
    for (int jj = 0; jj < RJ; ++jj) {
      if constexpr (PH == 7) {
        if (kvalid && (rowValid >> jj & 1)) {
          const int64_t o = ii * G.s0 + jrow[jj] * G.s1 + kc;
          const int c0 = A.gnd - 3;   // component of internal axis a is a - (3 - nd)
          if (act0) *reinterpret_cast<V*>(A.out + (int64_t)c0 * G.ncell + o) = g0r[jj];
          *reinterpret_cast<V*>(A.out + (int64_t)(c0 + 1) * G.ncell + o) = g1r[jj];
          *reinterpret_cast<V*>(A.out + (int64_t)(c0 + 2) * G.ncell + o) = g2r[jj];
        }
        continue;
      }
      V outd;   // phase A: d' ; phase B: new r
      V outx;   // phase B: new x
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const bool inS = iS && (rowS >> jj & 1) && (colS >> v & 1);
        const T xc = ec[jj][v];
        if (PH == 5 || PH == 6 || PH == 8) {
          // own cells: p' (or s) everywhere, v' = A p' (or t = A s) on the interior set
          const bool mine = kvalid && (rowValid >> jj & 1);
          const T an = inS ? res[jj][v] : (T)0;
          outd[v] = xc;
          outx[v] = an;
          const T r0c = xv[jj][v];
          if (PH == 5 || PH == 8) {
            T p = r0c * an;
            s0 += inS ? (double)p : 0.0;
          } else {
            T p = xc * xc;
            s0 += mine ? (double)p : 0.0;   // |s|^2 over every node (tol of linalg.py:233)
            T a = an * xc, b = an * an, cc = r0c * an;
            s1 += inS ? (double)a : 0.0;
            s2 += inS ? (double)b : 0.0;
            s3 += inS ? (double)cc : 0.0;
          }
        } else if (PH == 2) {
          outd[v] = (inS || !A.interior_only) ? res[jj][v] : (T)0;
        } else if (PH == 3) {
          outd[v] = inS ? res[jj][v] : xc;
        } else if (PH == 4) {
          const T xn = inS ? res[jj][v] : xc;
          const bool offshell = inS && !(iShell || (rowShell >> jj & 1) || (colShell >> v & 1));
          T df = xn - xc;
          T p2 = df * df;
          s1 += offshell ? (double)p2 : 0.0;
          outd[v] = xn;
        } else if (PH == 0) {
          const T e = inS ? xc : (T)0;
          outd[v] = e;
          T p = e * res[jj][v];
          s0 += (double)p;    // (off the interior set e is 0 and the product a signed zero: no second select)
        } else {
          const T xo = xv[jj][v];
          T ad = alpha * xc;
          T xn = xo + ad;
          T aAd = alpha * res[jj][v];
          T rn = rv[jj][v] - aAd;
          xn = inS ? xn : xo;
          rn = inS ? rn : (T)0;
          T p = rn * rn;
          s0 += (double)p;    // (rn is 0 off the interior set)
          const bool offshell = inS && !(iShell || (rowShell >> jj & 1) || (colShell >> v & 1));
          T df = xn - xo;
          T p2 = df * df;
          s1 += offshell ? (double)p2 : 0.0;
          outd[v] = rn;
          outx[v] = xn;
        }
      }
      if (kvalid && (rowValid >> jj & 1)) {
        const int64_t o = ii * G.s0 + jrow[jj] * G.s1 + kc;
        const int64_t ob = PITCH ? ii * fs0 + jrow[jj] * fs1 + kcf : o;   // BiCGSTAB: pitched outputs
        if (PH == 8) {   // p' is the input field itself (k_bicg_x formed it): only v' = A p' leaves
          *reinterpret_cast<V*>(A.out2 + ob) = outx;
        } else if (PH == 5 || PH == 6) {   // (non-temporal r0 loads / p, v, s, t stores: within the noise, A/B)
          // phase 6 on one GPU stores t alone (out = null): s = r - alpha v is re-formed, operation for operation, by the
          // x / r update that follows (k_bicg_x<..., SRV>), which reads r and v anyway -- 15 array passes for 16
          if (PH == 5 || A.out) *reinterpret_cast<V*>(A.out + ob) = outd;
          *reinterpret_cast<V*>(A.out2 + ob) = outx;
        } else if (PH >= 2) {
          *reinterpret_cast<V*>(A.out + o) = outd;
        } else if (PH == 0) {
          if (PITCH) __builtin_nontemporal_store(outd, reinterpret_cast<V*>(A.dnew + ii * fs0 + jrow[jj] * fs1 + kcf));
          else if (NARROW) *reinterpret_cast<V*>(A.dnew + o) = outd;
          else __builtin_nontemporal_store(outd, reinterpret_cast<V*>(A.dnew + o));  // -1.5 % (measured)
        } else {
          if (PITCH) {
            __builtin_nontemporal_store(outd, reinterpret_cast<V*>(A.rw_out + ii * fs0 + jrow[jj] * fs1 + kcf));
            T* xp = A.x + ii * G.s0 + jrow[jj] * G.s1;
#pragma unroll
            for (int v = 0; v < VEC; ++v)
              if (kg + v < G.n2) xp[kg + v] = outx[v];
          } else if (NARROW) {
            *reinterpret_cast<V*>(A.x + o) = outx;
            *reinterpret_cast<V*>(A.rw_out + o) = outd;
          } else {
            __builtin_nontemporal_store(outx, reinterpret_cast<V*>(A.x + o));
            __builtin_nontemporal_store(outd, reinterpret_cast<V*>(A.rw_out + o));
          }
          if (A.send_lo && ii == 0) *reinterpret_cast<V*>(A.send_lo + jrow[jj] * G.s1 + kc) = outd;
          if (A.send_hi && ii == G.n0 - 1) *reinterpret_cast<V*>(A.send_hi + jrow[jj] * G.s1 + kc) = outd;
        }
      }
    }

    // ---- rotate the register planes; plane m+2 becomes "ahead" (first use of its loads) --------
#pragma unroll
    for (int jj = 0; jj < RJ; ++jj) {
      ea[jj] = ec[jj];
      ec[jj] = eb[jj];
    }
    if (more) {
      finish_own(w, eb);
      finish_halo(w, hv, hs);
    }
    __syncthreads();
  }

  if (PH == 0) {
    double s[1] = {s0};
    pa_block_reduce_store<1>(s, A.partials);
  } else if (PH == 1 || PH == 4) {
    double s[2] = {s0, s1};
    pa_block_reduce_store<2>(s, A.partials);
  } else if (PH == 5 || PH == 8) {
    double s[1] = {s0};
    pa_block_reduce_store<1>(s, A.partials);
  } else if (PH == 6) {
    double s[4] = {s0, s1, s2, s3};
    pa_block_reduce_store<4>(s, A.partials);
  }
}

// ---- host side -------------------------------------------------------------------------------
// The equations the tiled kernels evaluate: one Laplacian term, or -- for A x and the BiCGSTAB phases --
// a Laplacian with a scalar coefficient plus a Div with a scalar advection speed, in either order
// (steady advection-diffusion).  il / id: positions of the two terms (id = -1: no Div).
template <typename T>
static bool eq_lap_div(const DevEq<T>& E, int& il, int& id, bool field_speed = false) {
  il = id = -1;
  if (E.nterms == 1 && E.t[0].kind == PA_OP_LAPLACIAN) { il = 0; return true; }
  if (E.nterms == 1) {  // the Div term alone (explicit Div of a scalar speed): il stays -1
    const int k = E.t[0].kind;
    // a speed FIELD only for the upwind schemes (they read u at the cell; central reads its neighbours too)
    const bool speed_ok = !E.t[0].u_f || (field_speed && k != PA_OP_DIV_CENTRAL);
    if ((k == PA_OP_DIV_CENTRAL || k == PA_OP_DIV_UPWIND_COMPAT || k == PA_OP_DIV_UPWIND) && speed_ok &&
        !E.t[0].coeff_f) { id = 0; return true; }
    return false;
  }
  if (E.nterms != 2) return false;
  for (int q = 0; q < 2; ++q) {
    const int k = E.t[q].kind;
    if (k == PA_OP_LAPLACIAN) il = q;
    else if ((k == PA_OP_DIV_CENTRAL || k == PA_OP_DIV_UPWIND_COMPAT || k == PA_OP_DIV_UPWIND) && !E.t[q].u_f) id = q;
  }
  return il >= 0 && id >= 0 && !E.t[il].coeff_f;
}

// 0: not for the tiled kernels; 1: 16-byte vector lanes; 2: one cell per lane (NARROW); 4: axisymmetric mesh -- the 2-D
// marching kernel k_cg2d<..., RZ> or nothing (only the callers that have a k_cg2d phase ask with rz_ok)
template <typename T>
static int cg3d_mode(const pa_ctx* c, const DevEq<T>& E, std::initializer_list<const void*> ptrs,
                     bool allow_div = false, bool field_speed = false, bool rz_ok = false) {
  if (!c->fastpath) return 0;
  if (c->coord != PA_COORD_XYZ) {   // r-dependent rows: k_cg2d where the phase has one, else the generic kernels
    if (!rz_ok || c->coord != PA_COORD_RZ || c->ndim != 2 || E.nterms != 1 || E.t[0].kind != PA_OP_LAPLACIAN || E.t[0].coeff_f)
      return 0;
    uintptr_t bits = 0;
    for (const void* q : ptrs) bits |= (uintptr_t)q;
    if ((bits & 15) || c->G.n2 % VecOf<T>::N != 0) return 0;   // (odd rows: only in the PITCH layout, mode 3 of the callers)
    return 4;
  }
  if (c->ndim != 3 && c->ndim != 2) return 0;
  int il, id;
  if (!eq_lap_div<T>(E, il, id, field_speed) || (id >= 0 && !allow_div)) return 0;
  if ((c->ndim == 3 && c->G.n0 < 3) || c->G.n1 < 3 || c->G.n2 < 3) return 0;
  uintptr_t bits = il >= 0 ? (uintptr_t)E.t[il].coeff_f : 0;
  if (id >= 0) bits |= (uintptr_t)E.t[id].u_f;
  for (const void* q : ptrs) bits |= (uintptr_t)q;
  if (bits & (sizeof(T) - 1)) return 0;
  constexpr int VEC = VecOf<T>::N;
  if ((bits & 15) || c->G.n2 % VEC != 0 || c->G.n2 < 2 * VEC) return 2;
  return 1;
}

static int cus_of(pa_ctx* c) {
  static int cus = 0;
  if (!cus) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

template <typename T, int RJ, int PHASE, bool CF, int KIND = 0, int LAY = 0>
static int blocks_per_cu() {
  static int cached = 0;
  if (!cached) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_cg3d<T, RJ, PHASE, CF, KIND, LAY>, 256, 0) != hipSuccess || n <= 0) n = 2;
    cached = n;
  }
  return cached;
}

template <typename T, int RJ, int PHASE, bool CF = false, int KIND = 0, int LAY = 0>
static int launch_cg3d(pa_ctx* c, Cg3dArgs<T>& A) {
  constexpr bool NARROW = LAY == 1;
  constexpr int VEC = NARROW ? 1 : VecOf<T>::N;
  constexpr int TJ = 4 * RJ, TK = 64 * VEC;
  const DevGeom& G = c->G;
  // The CG phases only ever change cells of the interior set: the last row / column of a non-periodic
  // axis is a boundary node whose d', r stay 0 and whose x is left alone, so no tile needs to cover
  // it.  For the 2^k + 1 extents node-based meshes like, that removes a whole extra tile row and tile
  // column (257^2 planes: 64 instead of 85 tiles of 16 x 64).  cg_begin zeroes both direction buffers.
  // The same holds for the BiCGSTAB phases (5, 6, 8): p, v, s, t, r are 0 outside the interior set for the whole
  // solve (bicg_run_t zeroes every buffer these phases write before the first iteration).
  int64_t n1e = G.n1, n2e = G.n2;
  if (PHASE == 0 || PHASE == 1 || PHASE == 5 || PHASE == 6 || PHASE == 8) {
    if (G.bct[3] != PA_BC_PERIODIC && n1e > 2) n1e -= 1;
    if (G.bct[5] != PA_BC_PERIODIC && n2e > 2) n2e -= 1;
  }
  A.tiles_j = (int)((n1e + TJ - 1) / TJ);
  A.tiles_k = (int)((n2e + TK - 1) / TK);
  const int tiles = A.tiles_j * A.tiles_k;
  const int capacity = cus_of(c) * blocks_per_cu<T, RJ, PHASE, CF, KIND, LAY>();
  int chunks = capacity / tiles;
  if (chunks < 1) chunks = 1;
  if (chunks > G.n0) chunks = (int)G.n0;
  A.chunks = chunks;
  const int nblk = tiles * chunks;
  if (nblk > PA_MAX_PARTIALS) return 0;
  static int dbg = -1;
  if (dbg < 0) dbg = getenv("PYAPES_HIP_DEBUG") ? 8 : 0;
  if (dbg > 0) {
    --dbg;
    fprintf(stderr, "[pyapes_hip] k_cg3d phase %c%s: tiles %dx%d chunks %d (CI ~%lld) blocks %d, %d blocks/CU x %d CUs\n",
            (char)('A' + PHASE), LAY == 1 ? " (narrow)" : (LAY == 2 ? " (pitched)" : ""), A.tiles_j, A.tiles_k, chunks, (long long)(G.n0 / chunks), nblk,
            blocks_per_cu<T, RJ, PHASE, CF, KIND, LAY>(), cus_of(c));
  }
  if (c->plan_only) return nblk;   // pa_cg_fold_plan: the grid this launch would use
  hipLaunchKernelGGL((k_cg3d<T, RJ, PHASE, CF, KIND, LAY>), dim3(nblk), dim3(256), 0, c->stream, A);
  return nblk;
}

// rows per thread.  Measured over 48^3 .. 512^3, long and flat boxes, fp64 and fp32: for
// the CG / A x / Jacobi / BiCGSTAB phases 4 rows (16-row tiles) are best or within 1.5 % of best
// everywhere and 1 row is never best (64^3: 20.9 vs 24.8 us / iteration, 192^3: 101 vs 117); 2 rows win
// by a hair exactly where 16-row tiles leave chunks shorter than 24 planes and 8-row tiles do not (256^3
// fp64).  The explicit Euler step is instruction-bound and keeps its own rule.
template <typename T, int PHASE = 0>
static int pick_rj(pa_ctx* c, bool narrow = false, bool cg_phase = false) {
  const int VEC = narrow ? 1 : VecOf<T>::N;
  const DevGeom& G = c->G;
  if (!G.act[0]) return 4;  // 2-D: one plane, nothing to march; the biggest tile has the least halo
  // the CG phases do not tile the last boundary row / column of a non-periodic axis (launch_cg3d): 257 nodes are
  // 256 cells to cover, and the rule must see the tiles that will really be launched
  const int64_t n1e = G.n1 - ((cg_phase && G.bct[3] != PA_BC_PERIODIC && G.n1 > 2) ? 1 : 0);
  const int64_t n2e = G.n2 - ((cg_phase && G.bct[5] != PA_BC_PERIODIC && G.n2 > 2) ? 1 : 0);
  const int64_t tk = (n2e + 64 * VEC - 1) / (64 * VEC);
  const int cap = cus_of(c) * 2;
  auto chunk_len = [&](int rj) {
    const int64_t tiles = ((n1e + 4 * rj - 1) / (4 * rj)) * tk;
    const int64_t chunks = cap / tiles > 0 ? cap / tiles : 1;
    return G.n0 / chunks;
  };
  if (PHASE == 3) {
    // more rows per thread amortise the per-plane bookkeeping, and the step tolerates shorter chunks
    // (256^3 fp32: RJ 2 48 us / step, RJ 1 54); below that 2 rows stay best (fp32 64^3 12.9 vs 14.1 us,
    // 128^3 19.9 vs 22.0, 192^3 34.3 vs 38.3; fp64 indifferent), 1 row never is
    return chunk_len(4) >= 12 ? 4 : 2;
  }
  if (G.n1 <= 4) return 1;
  if (G.n1 <= 8) return 2;
  return (chunk_len(4) < 24 && chunk_len(2) >= 24) ? 2 : 4;
}

template <typename T, int PHASE, int NARROW>   // NARROW = LAY of k_cg3d: 0 vector, 1 one cell per lane, 2 pitched
static int launch_any_w(pa_ctx* c, Cg3dArgs<T>& A) {
  constexpr bool CF_OK = (PHASE == 0 || PHASE == 1 || PHASE == 2 || PHASE == 4 || PHASE == 9);
  if (A.coeff_f) {  // tensor coefficient: separate instantiation, so the scalar-coefficient kernels stay lean
    if (!CF_OK) return 0;
    if constexpr (CF_OK) {
      switch (pick_rj<T>(c, NARROW == 1)) {
        case 1: return launch_cg3d<T, 1, PHASE, true, 0, NARROW>(c, A);
        case 2: return launch_cg3d<T, 2, PHASE, true, 0, NARROW>(c, A);
        default: return launch_cg3d<T, 4, PHASE, true, 0, NARROW>(c, A);
      }
    }
  }
  if constexpr (PHASE == 3) {  // one instantiation per Div scheme
    const int rj = pick_rj<T, 3>(c, NARROW == 1);
#define PA_EULER_CASE(K)                                                   \
    case K:                                                                \
      switch (rj) {                                                        \
        case 1: return launch_cg3d<T, 1, 3, false, K, NARROW>(c, A);       \
        case 2: return launch_cg3d<T, 2, 3, false, K, NARROW>(c, A);       \
        default: return launch_cg3d<T, 4, 3, false, K, NARROW>(c, A);      \
      }
    switch (A.kind) {
      PA_EULER_CASE(PA_OP_DIV_CENTRAL)
      PA_EULER_CASE(PA_OP_DIV_UPWIND_COMPAT)
      PA_EULER_CASE(PA_OP_DIV_UPWIND)
      default: return 0;
    }
#undef PA_EULER_CASE
  }
  if constexpr (PHASE == 2 || PHASE == 5 || PHASE == 6 || PHASE == 8) {
    if (A.kind != 0) {  // Laplacian + Div(scalar u): one instantiation per scheme, two or four rows per thread
      const bool four = pick_rj<T>(c, NARROW == 1) == 4;
#define PA_DIV_CASE(K)                                                                   \
      case K:                                                                            \
        return four ? launch_cg3d<T, 4, PHASE, false, K, NARROW>(c, A)                   \
                    : launch_cg3d<T, 2, PHASE, false, K, NARROW>(c, A);
      switch (A.kind) {
        PA_DIV_CASE(PA_OP_DIV_CENTRAL)
        PA_DIV_CASE(PA_OP_DIV_UPWIND_COMPAT)
        PA_DIV_CASE(PA_OP_DIV_UPWIND)
        default: return 0;
      }
#undef PA_DIV_CASE
    }
  }
  const int rj = pick_rj<T>(c, NARROW == 1, PHASE == 0 || PHASE == 1 || PHASE == 5 || PHASE == 6 || PHASE == 8);
  // (Round 4 tried 32-row tiles, RJ = 8, for the fp64 CG phases at 512^3 -- half the halo rows per cell, against phase
  // A's +5.7 % of re-read halo rows: 370 / 411 VGPRs instead of 223 / 237, i.e. ONE wave per SIMD, and phase A went from
  // 0.59 to 0.69 ms while phase B stayed at 0.92-0.94 (three interleaved pairs on one box).  The instantiation is gone.)
  switch (rj) {
    case 1: return launch_cg3d<T, 1, PHASE, false, 0, NARROW>(c, A);
    case 2: return launch_cg3d<T, 2, PHASE, false, 0, NARROW>(c, A);
    default: return launch_cg3d<T, 4, PHASE, false, 0, NARROW>(c, A);
  }
}

// the CG phases in the PITCH layout (a plain Laplacian only: the other instantiations do not exist)
template <typename T, int PHASE>
static int launch_pitched(pa_ctx* c, Cg3dArgs<T>& A) {
  static_assert(PHASE == 0 || PHASE == 1 || PHASE == 5 || PHASE == 6 || PHASE == 8, "PITCH: CG / BiCGSTAB phases");
  if (A.coeff_f || A.kind != 0) return 0;
  switch (pick_rj<T>(c, false, true)) {
    case 1: return launch_cg3d<T, 1, PHASE, false, 0, 2>(c, A);
    case 2: return launch_cg3d<T, 2, PHASE, false, 0, 2>(c, A);
    default: return launch_cg3d<T, 4, PHASE, false, 0, 2>(c, A);
  }
}

template <typename T, int PHASE>
static int launch_any(pa_ctx* c, Cg3dArgs<T>& A, int mode) {
  if constexpr (PHASE == 0 || PHASE == 1 || PHASE == 5 || PHASE == 6 || PHASE == 8) {
    if (mode == 3) return launch_pitched<T, PHASE>(c, A);
  }
  return mode == 2 ? launch_any_w<T, PHASE, 1>(c, A) : launch_any_w<T, PHASE, 0>(c, A);
}

template <typename T>
static void fill_common(pa_ctx* c, const DevEq<T>& E, Cg3dArgs<T>& A) {
  int il = 0, id = -1;
  (void)eq_lap_div<T>(E, il, id, true);
  A.lap_off = (il < 0 && id >= 0) ? 1 : 0;
  if (il < 0) il = 0;
  A.G = c->G;
  A.lap = E.lap;
  A.coeff = E.t[il].coeff;
  A.coeff_f = E.t[il].coeff_f;
  A.sign = E.t[il].sign;
  A.has_coeff = E.t[il].has_coeff;
  A.sc = c->sc;
  if (id >= 0) {  // Laplacian + Div: scheme, speed, sign of the Div term, and whether it is listed first
    for (int a = 0; a < 3; ++a) {
      T h = (T)c->dx[a];
      A.hh[a] = h;
      A.h2[a] = (T)2 * h;
      A.ih[a] = (T)1 / h;
    }
    A.kind = E.t[id].kind;
    A.u = E.t[id].u;
    A.p0 = E.t[id].sign;
    A.p1 = id < il ? (T)1 : (T)0;
  }
}

template <typename T>
static void fill_h(const pa_ctx* c, Cg3dArgs<T>& A) {
  for (int a = 0; a < 3; ++a) {
    T h = (T)c->dx[a];
    A.hh[a] = h;
    A.h2[a] = (T)2 * h;
    A.ih[a] = (T)1 / h;
  }
}

