// pa_resident.hip -- small meshes: the whole CG / Jacobi / BiCGSTAB solve in ONE cooperative launch, the fields
// resident in LDS (linalg.py:74-279 and the Jacobi of SURVEY a15, same arithmetic as the kernels of pa_solver.hip).
//
// On the meshes the reference's tests and demos run (128^2, 33^3, 64^3 ...) an iteration of the launch-per-
// phase loops costs ~10 us per dependent kernel whatever the kernel does (DESIGN "small meshes").  Here the
// mesh is cut into <= 128 boxes, one workgroup each, all co-resident (hipLaunchCooperativeKernel); the fields
// (CG: x, r and the direction d with a one-cell halo) live in the workgroup's LDS for the whole solve -- the
// set-up included: the solver scalars come from the kernel arguments, the start is BC-filled and the first
// residual formed in the kernel (two more grid-wide steps) -- and an iteration costs two grid-wide steps:
//   step 1: partial d.Ad -> mailbox, arrive, wait, every workgroup sums all partials in the same fixed
//           order (same bits everywhere) -> alpha
//   step 2: x, r update, ordered BC fill (literal, face after face, in LDS), boundary-shell stop-test
//           term; partial r.r and |dx|^2 and the box's outer layers of r -> mailbox, arrive, wait -> beta,
//           stop test; d' = r + beta d on the own cells and -- from the neighbours' r layers and the old
//           halo of d -- on the halo, bit for bit what the neighbour computes, so d is never exchanged.
// A grid-wide step is an agent-scope release add on one counter plus an acquire spin (2 us at 16 boxes,
// 3.2 us at 64, 5 us at 128: profiles/tools/gridbar.hip).  Every spin is bounded; a wait that gives up makes
// every workgroup return without storing anything, and the host runs the launch-per-phase loop instead.
// Jacobi: one step per sweep (the outer layers of x travel with the stop-test partial).  BiCGSTAB: three.
//
// Scope: one GPU, any term list pa_apply_terms knows (xyz or rz; tensor coefficients / speed fields since round 4: read
// from device memory by the general-equation build), mesh <= 128
// boxes of <= 4096 cells (and what fits the LDS: BiCGSTAB keeps six arrays); a periodic axis is never cut (the
// fill of its faces reads the far end of the axis), the box is its own neighbour there.  Everything else runs
// the launch-per-phase loops.
#include "pa_host.h"
#include "pa_scalar_steps.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>

#define RES_MAXG_LIMIT 256   // one workgroup per CU of the MI355X
// Default 128: measured in round 3 with a limit of 256 (boxes of <= 4096 cells up to 100^3), the grid-wide
// step of 150-256 workgroups eats what the saved launches give -- CG 80^3 22.4 us / iteration against 22.9
// launch-per-phase, 96^3 30.7 against 27.4, 1024^2 26.3 against 23.0; BiCGSTAB 64^3 (150 boxes) 40.8 against 41.8,
// Jacobi 96^3 23.8 against 24.4 -- so above 128 boxes the launch-per-phase loops stay.
#define RES_MAXG 128
#define RES_MAXBOX 4096
#define RES_NS 4            // partial sums per grid-wide step (row width of the mailbox; CG / Jacobi use 2)
#define RES_LDS_LIMIT (156 * 1024)

template <typename T>
struct ResFace {      // one BC face, in list order (linalg.py:295-297)
  int face;           // internal face 2 * axis + side; -1: unused slot
  int type;
  T sval;             // dirichlet value / neumann additive constant for scalar V
  const T* vals;      // per-node g or V (face layout of the whole mesh)
  T dxf, ndir;
};

template <typename T>
struct ResArgs {
  ResFace<T> f[6];
  T c43, c13, c23;
  int P[3];            // boxes per axis
  int bmax[3];         // largest box extent per axis
  int nface;           // largest box face (nodes)
  int bc_static;
  T omega;
  T* x;                // in: the start (BCs not yet filled); out: the iterate
  const T* rhs;        // the right-hand side (after pa_rhs_adjust)
  double tol;          // FDMSolverConfig tol / max_it: the kernel builds the solver scalars itself
  long long max_it;
  T* x_old_out;        // Field.VARo on request
  SolverScalars* sc;
  unsigned long long* counter;
  int* fail;           // set when a grid-wide wait timed out
  double* parts;       // [2][G][RES_NS]
  T* mail;             // [2][G][6][nface]
  unsigned spin_max;
  unsigned o_h, o_p1, o_p2, o_bcc, o_sh, o_meta, o_lists;   // LDS byte offsets: haloed array, two plain arrays, BC constants, shell, per-cell words, work lists
  unsigned o_h2, o_h3, o_p3;   // BiCGSTAB: two more haloed arrays, one more plain one
  unsigned o_rz;               // LEAN on an rz mesh: the box's rows of the r-dependent coefficient table, [3][bmax[1]]
  int rz_on;
};

template <typename T>
struct BoxView {   // an LDS array addressed with the node indices of the mesh
  T* p;
  int s0, s1, off;
  __device__ __forceinline__ T& operator()(int i, int j, int k) const { return p[i * s0 + j * s1 + k + off]; }
};
template <typename T>
struct BoxAcc {    // accessor of pa_apply_terms
  BoxView<T> v;
  __device__ __forceinline__ T at(const DevGeom&, int64_t i, int64_t j, int64_t k) const {
    return v.p[(int)i * v.s0 + (int)j * v.s1 + (int)k + v.off];
  }
};

// a[axis] with a run-time axis: selects, not an indexed read (which would put the array -- and the kernel -- on
// scratch memory)
template <typename V>
__device__ __forceinline__ V res_pick(const V (&v)[3], int a) { return a == 0 ? v[0] : (a == 1 ? v[1] : v[2]); }

struct ResSync {
  unsigned long long* counter;
  int* fail;
  unsigned long long step;
  unsigned G, spin_max;
};

// arrive + wait: everything this workgroup stored before is visible to every workgroup that returns
__device__ __forceinline__ bool res_grid_wait(ResSync& S) {
  __shared__ int ok;
  __syncthreads();
  if (S.G == 1) {   // the mesh is one box: the workgroup barrier is the whole step (what the box writes into its own
    S.step += 1;    // mailbox along a periodic axis is ordered by that barrier too)
    return true;
  }
  if (threadIdx.x == 0) {
    // The outcome of a step must be the SAME in every workgroup -- one that passes the last step stores its box of
    // x, one that gives up tells the host to redo the solve from the x it was handed -- so giving up is an update
    // of the counter itself: a compare-and-swap that sets RES_POISON, which only succeeds while the counter is
    // still below the step's target.  Either every arrival is in before anybody gives up (all pass) or the poison
    // is in before the last arrival (nobody passes, now or in any later step).
    const unsigned long long RES_POISON = 1ull << 62;
    // test hook (option res_spin 0, "every wait gives up"): workgroup 0 poisons the counter BEFORE it arrives, so
    // no workgroup can ever find the step complete -- without this the outcome would hang on whether all arrivals
    // happen to be in before the first look at the counter (seen: 8 workgroups passing every step of a solve)
    if (S.spin_max == 0 && blockIdx.x == 0)
      __hip_atomic_fetch_or(S.counter, RES_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(S.counter, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long target = (S.step + 1) * S.G;
    int good = 0;
    unsigned long long v = 0;
    // relaxed polls, ONE acquire fence after the last: an acquire load per poll would invalidate the caches of
    // the XCD on every round, under the feet of the workgroups that are still working
    for (unsigned spin = 0; spin < S.spin_max; ++spin) {
      v = __hip_atomic_load(S.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v & RES_POISON) break;
      if (v >= target) { good = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    if (!good && !(v & RES_POISON)) {
      v = __hip_atomic_load(S.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (!(v & RES_POISON)) {
        if (v >= target) { good = 1; break; }
        if (__hip_atomic_compare_exchange_strong(S.counter, &v, v | RES_POISON, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (!good) __hip_atomic_store(S.fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok = good;
  }
  __syncthreads();
  S.step += 1;
  return ok != 0;
}

// sum over the 64 lanes of a wave, the result in every lane: an inclusive scan inside each row of 16 lanes with
// DPP row shifts (lanes without a source add 0), then the four row totals read from lanes 15 / 31 / 47 / 63.
// Fixed order, no LDS: ~25 instructions against the six dependent ds_bpermute round trips of a shuffle reduction.
__device__ __forceinline__ double res_dpp_shr(double x, const int ctrl_sel) {
  const long long b = __double_as_longlong(x);
  int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
  // row_shr:n = 0x110 + n ; row_mask / bank_mask all, bound_ctrl off: out-of-row lanes receive `old` = 0
  if (ctrl_sel == 1) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, false); }
  if (ctrl_sel == 2) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x112, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x112, 0xf, 0xf, false); }
  if (ctrl_sel == 4) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xf, 0xf, false); }
  if (ctrl_sel == 8) { lo = __builtin_amdgcn_update_dpp(0, lo, 0x118, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x118, 0xf, 0xf, false); }
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double res_lane(double x, const int lane) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double res_wave_sum(double x) {
  x += res_dpp_shr(x, 1);
  x += res_dpp_shr(x, 2);
  x += res_dpp_shr(x, 4);
  x += res_dpp_shr(x, 8);
  return ((res_lane(x, 15) + res_lane(x, 31)) + res_lane(x, 47)) + res_lane(x, 63);
}

// grid-wide sums: v (per thread) -> out[RES_NS] in LDS, the same bits in every workgroup
template <int NU>
__device__ __forceinline__ bool res_allreduce(ResSync& S, double* parts, double (&v)[NU], double* out) {
  __shared__ double sm[RES_NS][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* row = parts + (size_t)(S.step & 1) * S.G * RES_NS;
#pragma unroll
  for (int s = 0; s < NU; ++s) {
    const double x = res_wave_sum(v[s]);
    if (lane == 0) sm[s][wave] = x;
  }
  __syncthreads();
  if (threadIdx.x < NU) {
    const int s = threadIdx.x;
    double x = sm[s][0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) x += sm[s][w];
    if (S.G == 1) out[s] = x;   // one box: its partial IS the sum (the general path would add zeros to it)
    else __hip_atomic_store(row + (size_t)blockIdx.x * RES_NS + s, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!res_grid_wait(S)) return false;
  if (S.G == 1) return true;   // (res_grid_wait's barrier has published out[])
  if (wave == 0) {
#pragma unroll
    for (int s = 0; s < NU; ++s) {
      double x = 0.0;
      for (unsigned g = lane; g < S.G; g += 64)
        x += __hip_atomic_load(row + (size_t)g * RES_NS + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      x = res_wave_sum(x);
      if (lane == 0) out[s] = x;
    }
  }
  __syncthreads();
  return true;
}

// per-cell word, built once (the thread -> cell mapping is fixed for the whole solve): where the cell lives in
// the haloed and in the plain arrays, whether it is in the interior set S, and the boundary-row case of each axis
#define RES_M_H(m) ((int)((m) & 0x1fffu))
#define RES_M_P(m) ((int)(((m) >> 13) & 0xfffu))
#define RES_M_S(m) (((m) >> 25) & 1u)
#define RES_M_RC(m, a) ((int)(((m) >> (26 + 2 * (a))) & 3u))
#define RES_M_SHELL(m) (((((m) >> 26) & ((m) >> 27)) & 0x15u) != 0u)   // some axis has row case 3

// SOLVER 0: CG (H = d with halo, P1 = x, P2 = r)   1: Jacobi (H = x with halo, P1 = x', P2 = rhs)
//        2: BiCGSTAB (H = r, then s, then the new r -- with halo; H2 = p, H3 = v with halo; P1 = x, P2 = r0, P3 = t)
// LEAN: the equation is one Laplacian term with a scalar coefficient on an xyz mesh (the stencil is evaluated from
// the per-cell word and fixed LDS offsets); else pa_apply_terms on the box (any term list, rz)
template <typename T, int SOLVER, bool LEAN, int NT>
__global__ void __launch_bounds__(NT) k_resident(DevGeom G, DevEq<T> E, ResArgs<T> A) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ SolverScalars sc;
  __shared__ double red[RES_NS];
  const int tid = threadIdx.x;

  // ---- this workgroup's box (plain scalars and selects: an aggregate indexed with a run-time axis would put
  //      the kernel on scratch memory) -----------------------------------------------------------------
  int p0, p1, p2;
  {
    int g = blockIdx.x;
    p2 = g % A.P[2]; g /= A.P[2];
    p1 = g % A.P[1];
    p0 = g / A.P[1];
  }
  const int lo0 = (int)((int64_t)p0 * G.n0 / A.P[0]), b0 = (int)((int64_t)(p0 + 1) * G.n0 / A.P[0]) - lo0;
  const int lo1 = (int)((int64_t)p1 * G.n1 / A.P[1]), b1 = (int)((int64_t)(p1 + 1) * G.n1 / A.P[1]) - lo1;
  const int lo2 = (int)((int64_t)p2 * G.n2 / A.P[2]), b2 = (int)((int64_t)(p2 + 1) * G.n2 / A.P[2]) - lo2;
  const int nbox = b0 * b1 * b2;
  auto LO = [&](int a) -> int { return a == 0 ? lo0 : (a == 1 ? lo1 : lo2); };
  auto BB = [&](int a) -> int { return a == 0 ? b0 : (a == 1 ? b1 : b2); };
  auto PP = [&](int a) -> int { return a == 0 ? p0 : (a == 1 ? p1 : p2); };
  // own cell c -> box coordinates: two divisions by magic numbers (cells < 4096, divisors <= 4096: exact)
  const unsigned mp = (1u << 24) / (unsigned)(b1 * b2) + 1u, mk = (1u << 24) / (unsigned)b2 + 1u;
  auto decode = [&](int c, int& bi, int& bj, int& bk) {
    const unsigned plane = (unsigned)(b1 * b2);
    bi = (int)(((unsigned long long)(unsigned)c * mp) >> 24);
    const unsigned rem = (unsigned)c - (unsigned)bi * plane;
    bj = (int)(((unsigned long long)rem * mk) >> 24);
    bk = (int)(rem - (unsigned)bj * (unsigned)b2);
  };
  // nodes of box face 2 * a + side: q = u * nv + v with (u, v) the two other axes in ascending order
  auto face_dims = [&](int a, int& nu, int& nv) {
    nu = a == 0 ? b1 : b0;
    nv = a == 2 ? b1 : b2;
  };
  auto face_node = [&](int a, int pos, int q, int nv, int& i, int& j, int& k) {
    const int u = q / nv, v = q - u * nv;
    if (a == 0) { i = pos; j = lo1 + u; k = lo2 + v; }
    else if (a == 1) { i = lo0 + u; j = pos; k = lo2 + v; }
    else { i = lo0 + u; j = lo1 + v; k = pos; }
  };
  int h[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) h[a] = G.act[a] ? 1 : 0;
  // views: strides from the largest box, so that the LDS layout is the same in every workgroup
  BoxView<T> H, P1, P2;
  H.p = (T*)(smem + A.o_h);
  H.s1 = A.bmax[2] + 2 * h[2];
  H.s0 = (A.bmax[1] + 2 * h[1]) * H.s1;
  H.off = (h[0] - LO(0)) * H.s0 + (h[1] - LO(1)) * H.s1 + (h[2] - LO(2));
  P1.p = (T*)(smem + A.o_p1);
  P1.s1 = A.bmax[2];
  P1.s0 = A.bmax[1] * A.bmax[2];
  P1.off = -LO(0) * P1.s0 - LO(1) * P1.s1 - LO(2);
  P2 = P1;
  P2.p = (T*)(smem + A.o_p2);
  T* bcc = (T*)(smem + A.o_bcc);    // [6 list slots][nface]
  T* shold = (T*)(smem + A.o_sh);   // [6 faces][nface]
  unsigned* meta = (unsigned*)(smem + A.o_meta);
  // LEAN on an axisymmetric mesh: the r-dependent rows (cP, cM, cB of pa_apply_terms; tools.py:86-107) of this box's
  // r range, read once from the table of pa_coord_set
  T* const rzl = (T*)(smem + A.o_rz);
  const int rzs = A.bmax[1];
  const bool rz_on = LEAN && A.rz_on != 0;
  // work lists of the per-iteration face loops, dense by (face, q) with uniform bases (built once below):
  //   BC fill: index of the face node in X, in list order per BC slot (bcc in the same order)
  //   shell:   index in X of the shell nodes this box owns (0xffff: owned by another face), shold in the same order
  //   publish: index of the layer cell in the source array + its mailbox offset
  //   receive: index of the halo cell in H + the offset of its value in the neighbour's mailbox (+ in-S flag)
  unsigned short* bcD = (unsigned short*)(smem + A.o_lists);
  unsigned short* shD = bcD + 6 * A.nface;
  unsigned short* pubS = shD + 6 * A.nface;
  unsigned short* pubO = pubS + 6 * A.nface;
  unsigned short* rcvH = pubO + 6 * A.nface;
  unsigned* rcvO = (unsigned*)(smem + A.o_lists + (size_t)10 * 6 * A.nface);
  __shared__ int fm_cnt[6], fm_base[6], fm_sst[6], fm_n[6];   // BC slot w: nodes of this box, list base, signed stride to prev, axis length
  __shared__ int n_sh, n_sh_all, n_pub, n_rcv;
  const BoxView<T>& X = SOLVER == 1 ? H : P1;   // the iterate (BC fill, shell term)
  BoxView<T> H2 = H, H3 = H, P3 = P1;
  if (SOLVER == 2) {
    H2.p = (T*)(smem + A.o_h2);
    H3.p = (T*)(smem + A.o_h3);
    P3.p = (T*)(smem + A.o_p3);
  }
  T* const H2p = H2.p;
  T* const H3p = H3.p;
  T* const P3p = P3.p;
  T* const Hp = H.p;
  T* const P1p = P1.p;
  T* const P2p = P2.p;
  const int hs0 = H.s0, hs1 = H.s1;

  ResSync S;
  S.counter = A.counter; S.fail = A.fail; S.step = 0; S.G = gridDim.x; S.spin_max = A.spin_max;

  if (tid == 0) {   // init_scalars of pa_solver.hip (linalg.py:90, 109, 201-204)
    sc = SolverScalars{};
    sc.tolerance = A.tol;
    sc.max_it = A.max_it;
    sc.tol = 1.0;
    sc.rho = 1.0; sc.alpha = 1.0; sc.omega = 1.0;
    sc.done = (SOLVER != 2 && !(1.0 > A.tol)) ? 1 : 0;   // `while tol > tolerance` with tol = 1.0; BiCGSTAB: `while not finished`
  }
  __syncthreads();

  // neighbour boxes: workgroup index, or -1 at the ends of the mesh
  int nb[6];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int st = a == 0 ? A.P[1] * A.P[2] : (a == 1 ? A.P[2] : 1);
    nb[2 * a] = PP(a) > 0 ? (int)blockIdx.x - st : -1;
    nb[2 * a + 1] = PP(a) + 1 < A.P[a] ? (int)blockIdx.x + st : -1;
    // a periodic axis is never cut (the fill of its faces reads the far end of the axis): the box is its own
    // neighbour there, the wrap-around layers travel through its own mailbox like any other layer
    if (G.act[a] && G.bct[2 * a] == 4) nb[2 * a] = nb[2 * a + 1] = (int)blockIdx.x;
  }

  // ---- load ---------------------------------------------------------------------------------------
  if (rz_on) {
    for (int j = tid; j < b1; j += NT) {
      rzl[j] = E.rz[lo1 + j];
      rzl[rzs + j] = E.rz[E.rz_n + lo1 + j];
      rzl[2 * rzs + j] = E.rz[2 * E.rz_n + lo1 + j];
    }
  }
  for (int c = tid; c < nbox; c += NT) {
    int bi, bj, bk;
    decode(c, bi, bj, bk);
    const int i = LO(0) + bi, j = LO(1) + bj, k = LO(2) + bk;
    const int64_t o = (int64_t)i * G.s0 + (int64_t)j * G.s1 + k;
    const int hidx = i * H.s0 + j * H.s1 + k + H.off;
    const int pidx = i * P1.s0 + j * P1.s1 + k + P1.off;
    unsigned m = (unsigned)hidx | ((unsigned)pidx << 13);
    if (pa_in_S(G, i, j, k)) m |= 1u << 25;
    // row case per axis: 0 plain, 1 / 2 the lower / upper neumann | symmetry row, 3 = the node is a boundary node
    // of the axis (a plain row; only a periodic face puts such a node into S, and its |dx|^2 is the shell term's)
    auto rowc = [&](int a, int g, int64_t N) -> unsigned {
      const int rc = pa_row_case(G, a, g, N, G.treat);
      return (rc == 0 && G.act[a] && (g == 0 || g == N - 1)) ? 3u : (unsigned)rc;
    };
    m |= rowc(0, i, G.n0) << 26;
    m |= rowc(1, j, G.n1) << 28;
    m |= rowc(2, k, G.n2) << 30;
    meta[c] = m;
    // the start as the caller hands it over (BCs not yet filled) and the right-hand side; everything else is
    // built below, in the kernel: BC fill, residual, its norm (what pa_cg_begin does with five launches)
    if (SOLVER == 1) Hp[hidx] = A.x[o]; else P1p[pidx] = A.x[o];
    P2p[pidx] = A.rhs[o];
    if (SOLVER == 2) {
      H2p[hidx] = (T)0;       // p = v = 0 (linalg.py:203-204)
      H3p[hidx] = (T)0;
    }
  }
  if (SOLVER == 2) {   // halo of p and v: zero like the fields
#pragma unroll
    for (int dir = 0; dir < 6; ++dir) {
      const int a = dir >> 1, side = dir & 1;
      if (nb[dir] < 0) continue;
      int nu, nv;
      face_dims(a, nu, nv);
      const int pos = side == 0 ? LO(a) - 1 : LO(a) + BB(a);
      for (int q = tid; q < nu * nv; q += NT) {
        int i, j, k;
        face_node(a, pos, q, nv, i, j, k);
        H2(i, j, k) = (T)0;
        H3(i, j, k) = (T)0;
      }
    }
  }
  const int64_t Ng[3] = {G.n0, G.n1, G.n2};
  auto touches = [&](int f) -> bool {
    const int a = f >> 1;
    const int pa = PP(a);
    return res_pick(G.act, a) && ((f & 1) == 0 ? pa == 0 : pa == res_pick(A.P, a) - 1);
  };
  // ---- work lists (see above) and the BC constants, by list slot --------------------------------------
  T* const Xp = X.p;
  {
    int base = 0;
#pragma unroll
    for (int w = 0; w < 6; ++w) {
      const ResFace<T>& F = A.f[w];
      int cnt = 0;
      if (F.face >= 0 && touches(F.face)) {
        const int a = F.face >> 1, side = F.face & 1;
        int nu, nv;
        face_dims(a, nu, nv);
        cnt = nu * nv;
        const int64_t gnv = a == 2 ? G.n1 : G.n2;
        const int pos = side == 0 ? 0 : (int)res_pick(Ng, a) - 1;
        for (int q = tid; q < cnt; q += NT) {
          int i, j, k;
          face_node(a, pos, q, nv, i, j, k);
          const int u = q / nv, v = q - u * nv;
          const int64_t gq = (int64_t)((a == 0 ? LO(1) : LO(0)) + u) * gnv + ((a == 2 ? LO(1) : LO(2)) + v);
          T ct = F.sval;
          if (F.vals) {
            ct = F.vals[gq];
            if (F.type == 2) {
              ct = A.c23 * ct;
              ct = ct * F.dxf;
              ct = ct * F.ndir;
            }
          }
          bcc[base + q] = ct;
          bcD[base + q] = (unsigned short)(i * X.s0 + j * X.s1 + k + X.off);
        }
        if (tid == 0) {
          const int st = a == 0 ? X.s0 : (a == 1 ? X.s1 : 1);
          fm_sst[w] = side == 0 ? st : -st;
          fm_n[w] = (int)res_pick(Ng, a);
        }
      }
      if (tid == 0) { fm_cnt[w] = cnt; fm_base[w] = base; }
      base += cnt;
    }
  }
  __syncthreads();
  {
    int base = 0;
#pragma unroll
    for (int f = 0; f < 6; ++f) {
      if (!touches(f)) continue;
      const int a = f >> 1;
      int nu, nv;
      face_dims(a, nu, nv);
      const int pos = (f & 1) == 0 ? 0 : (int)Ng[a] - 1;
      for (int q = tid; q < nu * nv; q += NT) {
        int i, j, k;
        face_node(a, pos, q, nv, i, j, k);
        bool owned = true;   // each shell node once: the rule of pa_shell_node
        if (a >= 1 && G.act[0] && (i == 0 || i == G.n0 - 1)) owned = false;
        if (a == 2 && G.act[1] && (j == 0 || j == G.n1 - 1)) owned = false;
        shD[base + q] = owned ? (unsigned short)(i * X.s0 + j * X.s1 + k + X.off) : (unsigned short)0xffff;
      }
      base += nu * nv;
    }
    if (tid == 0) {
      n_sh_all = base;
      n_sh = (A.bc_static || SOLVER == 2) ? 0 : base;   // (BiCGSTAB stops on the residual: no shell term)
    }
  }
  {
    const BoxView<T>& SRC = SOLVER == 0 ? P2 : H;   // what travels: CG the residual, Jacobi the iterate (BiCGSTAB: haloed arrays)
    int base = 0;
#pragma unroll
    for (int dir = 0; dir < 6; ++dir) {
      if (nb[dir] < 0) continue;
      const int a = dir >> 1;
      int nu, nv;
      face_dims(a, nu, nv);
      const int pos_out = (dir & 1) == 0 ? LO(a) : LO(a) + BB(a) - 1;
      const int pos_in = (dir & 1) == 0 ? LO(a) - 1 : LO(a) + BB(a);
      for (int q = tid; q < nu * nv; q += NT) {
        int i, j, k;
        face_node(a, pos_out, q, nv, i, j, k);
        pubS[base + q] = (unsigned short)(i * SRC.s0 + j * SRC.s1 + k + SRC.off);
        pubO[base + q] = (unsigned short)(dir * A.nface + q);
        face_node(a, pos_in, q, nv, i, j, k);
        rcvH[base + q] = (unsigned short)(i * H.s0 + j * H.s1 + k + H.off);
        rcvO[base + q] = (unsigned)((nb[dir] * 6 + (dir ^ 1)) * A.nface + q) | (pa_in_S(G, pa_wrap(i, G.n0), pa_wrap(j, G.n1), pa_wrap(k, G.n2)) ? 0x80000000u : 0u);
      }
      base += nu * nv;
    }
    if (tid == 0) { n_pub = base; n_rcv = base; }
  }
  __syncthreads();

  // A(field in H) at own cell c: one Laplacian term from the per-cell word (the arithmetic of pa_apply_terms,
  // kind 0, rounding for rounding), or pa_apply_terms itself
  const T lsign = E.t[0].sign, lcoeff = E.t[0].coeff;
  const int lhas = E.t[0].has_coeff;
  auto stencil = [&](const BoxView<T>& SV, int c, unsigned m, T xc) -> T {
    T* const Hp = SV.p;   // (the source array of this evaluation)
    if (LEAN) {
      const int hx = RES_M_H(m);
      T ax = (T)0;
      T rP = E.lap.inv[1], rM = E.lap.inv[1], rB = E.lap.c23[1];   // axis 1: r on an rz mesh
      if (rz_on) {
        int bi, bj, bk;
        decode(c, bi, bj, bk);
        rP = rzl[bj];
        rM = rzl[rzs + bj];
        rB = rzl[2 * rzs + bj];
      }
      if ((m >> 26) == 0u) {   // no boundary row on any axis (almost every cell): the plain (1, -2, 1) / h^2 rows
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          if (!G.act[a]) continue;
          const int st = a == 0 ? hs0 : (a == 1 ? hs1 : 1);
          const T xp = Hp[hx + st], xm = Hp[hx - st];
          T s = (a == 1 ? rP : E.lap.inv[a]) * xp;
          T mm = E.lap.m2inv[a] * xc;
          s = s + mm;
          mm = (a == 1 ? rM : E.lap.inv[a]) * xm;
          s = s + mm;
          ax = ax + s;
        }
        if (lhas) ax = ax * lcoeff;
        ax = ax * lsign;
        return (T)0 + ax;
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        if (!G.act[a]) continue;
        const int st = a == 0 ? hs0 : (a == 1 ? hs1 : 1);
        const int rc = RES_M_RC(m, a);
        const T cB = a == 1 ? rB : E.lap.c23[a];
        T cP = a == 1 ? rP : E.lap.inv[a], cC = E.lap.m2inv[a], cM = a == 1 ? rM : E.lap.inv[a];
        if (rc == 1) { cP = cB; cC = -cB; cM = (T)0; }
        if (rc == 2) { cP = (T)0; cC = -cB; cM = cB; }
        const T xp = Hp[hx + st], xm = Hp[hx - st];
        T s = cP * xp;
        T mm = cC * xc;
        s = s + mm;
        mm = cM * xm;
        s = s + mm;
        ax = ax + s;
      }
      if (lhas) ax = ax * lcoeff;
      ax = ax * lsign;
      return (T)0 + ax;
    } else {
      int bi, bj, bk;
      decode(c, bi, bj, bk);
      const BoxAcc<T> acc{SV};
      return pa_apply_terms<T>(G, E, acc, LO(0) + bi, LO(1) + bj, LO(2) + bk, xc);
    }
  };

  // the ordered BC fill of the box's part of the shell (literal: face after face, a later face reads what an
  // earlier one wrote), then sum (x_new - x_old)^2 over the shell nodes this box owns
  auto bc_fill = [&]() {
#pragma unroll
    for (int w = 0; w < 6; ++w) {
      const int cnt = fm_cnt[w];
      if (cnt > 0) {
        const int base = fm_base[w], sst = fm_sst[w], type = A.f[w].type;
        for (int n = tid; n < cnt; n += NT) {
          const int d = bcD[base + n];
          T val;
          if (type == 1) {
            val = bcc[base + n];
          } else if (type == 2) {
            T t1 = A.c43 * Xp[d + sst];
            T t2 = A.c13 * Xp[d + 2 * sst];
            t1 = t1 - t2;
            val = t1 + bcc[base + n];
          } else if (type == 3) {
            val = Xp[d + sst];
          } else if (sst > 0) {   // periodic lower: x[0] = x[1] - x[N-1] + x[N-2] (bcs.py:245-262)
            const int N = fm_n[w];
            T t1 = Xp[d + sst] - Xp[d + (N - 1) * sst];
            val = t1 + Xp[d + (N - 2) * sst];
          } else {                // periodic upper: x[N-1] = x[0], which the lower face (earlier in the list or not) has set
            val = Xp[d + (fm_n[w] - 1) * sst];
          }
          Xp[d] = val;
        }
        __syncthreads();
      }
    }
  };
  auto bc_fill_and_shell = [&](double& acc) {
    bc_fill();
    const int ns = n_sh;
    for (int n = tid; n < ns; n += NT) {
      const int d = shD[n];
      if (d == 0xffff) continue;
      const T xn = Xp[d];
      T df = xn - shold[n];
      T p = df * df;
      acc += (double)p;
      shold[n] = xn;
    }
  };

  // the box's outer layers -> this workgroup's mailboxes (one per direction with a neighbour)
  auto publish = [&](const T* SRCp) {
    T* mine = A.mail + ((size_t)(S.step & 1) * gridDim.x + blockIdx.x) * 6 * A.nface;
    const int np = n_pub;
    for (int n = tid; n < np; n += NT)
      __hip_atomic_store(mine + pubO[n], SRCp[pubS[n]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // the neighbours' layers of the step just completed -> the halo of DST
  auto receive = [&](T* DSTp) {
    const T* theirs = A.mail + (size_t)((S.step - 1) & 1) * gridDim.x * 6 * A.nface;
    const int nr = n_rcv;
    for (int n = tid; n < nr; n += NT)
      DSTp[rcvH[n]] = __hip_atomic_load(theirs + (rcvO[n] & 0x7fffffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  // ---- set-up, in the kernel: what pa_cg_begin / the heads of pa_jacobi and pa_bicgstab do with a BC-fill launch,
  //      the A x and residual kernels and a single-block reduction (linalg.py:97-107, 196-212) ----------------
  bool timed_out = false, entry_done = false;
  do {
    bc_fill();                                   // x <- B(x) (linalg.py:97), literal like every later fill
    __syncthreads();
    {
      const int ns = n_sh_all;                   // the filled shell as x_old (k_shell, mode 0)
      for (int n = tid; n < ns; n += NT) {
        const int d = shD[n];
        if (d != 0xffff) shold[n] = Xp[d];
      }
    }
    if (SOLVER != 1)                             // x with a halo, for the first residual (H becomes d / r below)
      for (int c = tid; c < nbox; c += NT) {
        const unsigned m = meta[c];
        Hp[RES_M_H(m)] = P1p[RES_M_P(m)];
      }
    __syncthreads();
    if (sc.done) { entry_done = true; break; }   // `while tol > tolerance` false at entry: only the fill happened
    publish(SOLVER == 0 ? P1p : Hp);             // the neighbours' outer layers of the FILLED x
    double z[1] = {0.0};
    if (!res_allreduce(S, A.parts, z, red)) { timed_out = true; break; }
    receive(Hp);
    __syncthreads();
    if (SOLVER == 1) break;
    // r = (b - A x) on S, 0 elsewhere ; sum r.r (k_cg_init) ; d = r (CG) ; r0 = r, p = v = 0 (BiCGSTAB)
    double v1[1] = {0.0};
    for (int c = tid; c < nbox; c += NT) {
      const unsigned m = meta[c];
      const int px = RES_M_P(m);
      T rv = (T)0;
      if (RES_M_S(m)) {
        const T ax = stencil(H, c, m, Hp[RES_M_H(m)]);
        rv = P2p[px] - ax;
        T p = rv * rv;
        v1[0] += (double)p;
      }
      P2p[px] = rv;
    }
    __syncthreads();
    for (int c = tid; c < nbox; c += NT) {
      const unsigned m = meta[c];
      Hp[RES_M_H(m)] = P2p[RES_M_P(m)];
    }
    __syncthreads();
    publish(SOLVER == 0 ? P2p : Hp);
    if (!res_allreduce(S, A.parts, v1, red)) { timed_out = true; break; }
    receive(Hp);
    if (tid == 0) {
      sc.rr = (double)(T)red[0];                 // k_cg_post_init
      if (SOLVER == 2) {                         // linalg.py:201-212: rho' = r0.r0, tol0, the first beta = rho' / 1 * 1 / 1
        sc.rho_next = sc.rr;
        sc.tol = (double)(T)sqrt((T)sc.rr);
        T b = (T)sc.rho_next / (T)1.0;
        b = b * (T)1.0;
        b = b / (T)1.0;
        sc.beta = (double)b;
        sc.rho = sc.rho_next;
      }
    }
    __syncthreads();
  } while (0);

#ifdef PA_RES_TIMING   // measurement build (PA_EXTRA_FLAGS=-DPA_RES_TIMING): where an iteration's time goes, workgroup 0
  unsigned long long tt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tl = wall_clock64();
  long long tn = 0;
#define RES_T(q) do { const unsigned long long t_ = wall_clock64(); tt[q] += t_ - tl; tl = t_; } while (0)
#else
#define RES_T(q) do { } while (0)
#endif
  for (; !timed_out && !entry_done;) {
    if (A.x_old_out) {   // Field.VARo: the iterate before this iteration's update
      for (int c = tid; c < nbox; c += NT) {
        int bi, bj, bk;
        decode(c, bi, bj, bk);
        const int i = LO(0) + bi, j = LO(1) + bj, k = LO(2) + bk;
        A.x_old_out[(int64_t)i * G.s0 + (int64_t)j * G.s1 + k] = X(i, j, k);
      }
    }
    if (SOLVER == 0) {
      // ---- Ad = A(d) on S, alpha = r.r / d.Ad (linalg.py:115-120) ----------------------------------
      double v[2] = {0.0, 0.0};
      for (int c = tid; c < nbox; c += NT) {
        const unsigned m = meta[c];
        if (RES_M_S(m)) {
          const T e = Hp[RES_M_H(m)];
          const T Ad = stencil(H, c, m, e);
          T p = e * Ad;
          v[0] += (double)p;
        }
      }
      RES_T(0);   // pass A
      if (!res_allreduce(S, A.parts, v, red)) { timed_out = true; break; }
      RES_T(1);   // step 1
      if (tid == 0) pa_logic_a<T>(&sc, red);
      __syncthreads();
      RES_T(2);   // alpha
      const T alpha = (T)sc.alpha;
      // ---- x += alpha d ; r -= alpha A d ; BC fill ; partial r.r, |dx|^2 (linalg.py:122-134) -------
      v[0] = 0.0; v[1] = 0.0;
      for (int c = tid; c < nbox; c += NT) {
        const unsigned m = meta[c];
        if (RES_M_S(m)) {
          const int px = RES_M_P(m);
          const T dc = Hp[RES_M_H(m)];
          const T Ad = stencil(H, c, m, dc);
          const T xo = P1p[px];
          T ad = alpha * dc;
          T xn = xo + ad;
          P1p[px] = xn;
          T aAd = alpha * Ad;
          T rn = P2p[px] - aAd;
          P2p[px] = rn;
          T p = rn * rn;
          v[0] += (double)p;
          if (!RES_M_SHELL(m)) {
            T df = xn - xo;
            T p2 = df * df;
            v[1] += (double)p2;
          }
        }
      }
      __syncthreads();
      RES_T(3);   // pass B
      if (!A.bc_static) bc_fill_and_shell(v[1]);
      RES_T(4);   // BC fill + shell
      publish(P2p);
      RES_T(5);   // publish
      // sums: red[0] = r.r, red[1] = |x_new - x_old|^2 ; pa_logic_b reads them as sums[1], sums[2]
      if (!res_allreduce(S, A.parts, v, red)) { timed_out = true; break; }
      RES_T(6);   // step 2
      if (tid == 0) {
        const double sums[3] = {0.0, red[0], red[1]};
        pa_logic_b<T>(&sc, sums);
      }
      __syncthreads();
      RES_T(7);   // beta, stop test
      if (sc.done) break;
      // ---- d' = r + beta d (linalg.py:141): own cells, then the halo from the neighbours' r layers ---
      const T beta = (T)sc.beta;
      for (int c = tid; c < nbox; c += NT) {
        const unsigned m = meta[c];
        const int hx = RES_M_H(m);
        T e = (T)0;
        if (RES_M_S(m)) {
          T bd = beta * Hp[hx];
          e = P2p[RES_M_P(m)] + bd;
        }
        Hp[hx] = e;
      }
      {
        const T* theirs = A.mail + (size_t)((S.step - 1) & 1) * gridDim.x * 6 * A.nface;   // the step just completed
        const int nr = n_rcv;
        for (int n = tid; n < nr; n += NT) {
          const unsigned o = rcvO[n];
          const int hx = rcvH[n];
          const T rv = __hip_atomic_load(theirs + (o & 0x7fffffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          T e = (T)0;
          if (o >> 31) {
            T bd = beta * Hp[hx];
            e = rv + bd;
          }
          Hp[hx] = e;
        }
      }
      __syncthreads();
      RES_T(8);   // d' own + halo
#ifdef PA_RES_TIMING
      ++tn;
#endif
    } else if (SOLVER == 2) {
      // ---- BiCGSTAB (linalg.py:162-279; the arithmetic of k_bicg_pv / _s / _t / _x and k_bicg_post) --------
      // p' = r + beta (p - omega v) on every cell, own and halo (the halos hold r, p and v of the neighbours)
      {
        const T beta = (T)sc.beta, omega = (T)sc.omega;
        auto pnew = [&](int hx) {
          T t = omega * H3p[hx];
          t = H2p[hx] - t;
          t = beta * t;
          H2p[hx] = Hp[hx] + t;
        };
        for (int c = tid; c < nbox; c += NT) pnew(RES_M_H(meta[c]));
        const int nr = n_rcv;
        for (int n = tid; n < nr; n += NT) pnew(rcvH[n]);
      }
      __syncthreads();
      // v' = A p' on S ; r0 . v' -> alpha (k_bicg_post stage 0)
      double v1[1] = {0.0};
      for (int c = tid; c < nbox; c += NT) {
        const unsigned m = meta[c];
        const int hx = RES_M_H(m);
        T vn = (T)0;
        if (RES_M_S(m)) {
          vn = stencil(H2, c, m, H2p[hx]);
          T p = P2p[RES_M_P(m)] * vn;
          v1[0] += (double)p;
        }
        H3p[hx] = vn;
      }
      __syncthreads();
      publish(H3p);
      if (!res_allreduce(S, A.parts, v1, red)) { timed_out = true; break; }
      if (tid == 0) {
        sc.itr += 1;
        T r0v = (T)red[0];
        T rho = (T)sc.rho;
        sc.alpha = pa_nan_to_num<T>(rho / r0v);
      }
      receive(H3p);
      __syncthreads();
      // s = r - alpha v' on every cell, own and halo, in place of r ; |s|^2
      const T alpha = (T)sc.alpha;
      double v4[4] = {0.0, 0.0, 0.0, 0.0};
      for (int c = tid; c < nbox; c += NT) {
        const int hx = RES_M_H(meta[c]);
        T av = alpha * H3p[hx];
        T sv = Hp[hx] - av;
        Hp[hx] = sv;
        T p = sv * sv;
        v4[0] += (double)p;
      }
      {
        const int nr = n_rcv;
        for (int n = tid; n < nr; n += NT) {
          const int hx = rcvH[n];
          T av = alpha * H3p[hx];
          Hp[hx] = Hp[hx] - av;
        }
      }
      __syncthreads();
      // t = A s on S ; t.s, t.t, r0.t -> stop test 1, omega, rho' (k_bicg_post stage 12)
      for (int c = tid; c < nbox; c += NT) {
        const unsigned m = meta[c];
        T tv = (T)0;
        if (RES_M_S(m)) {
          const T sc_ = Hp[RES_M_H(m)];
          tv = stencil(H, c, m, sc_);
          T a = tv * sc_;
          T b = tv * tv;
          T c3 = P2p[RES_M_P(m)] * tv;
          v4[1] += (double)a;
          v4[2] += (double)b;
          v4[3] += (double)c3;
        }
        P3p[RES_M_P(m)] = tv;
      }
      if (!res_allreduce(S, A.parts, v4, red)) { timed_out = true; break; }
      if (tid == 0) {
        T tol = (T)sqrt(red[0]);
        sc.tol = (double)tol;
        if (isnan(tol) || isinf(tol)) {
          sc.err = 1;
          sc.done = 1;
        } else {
          sc.finished_early = (sc.tol <= sc.tolerance) ? 1 : 0;
          if (!sc.finished_early) {
            T om = (T)pa_nan_to_num<T>((T)red[1] / (T)red[2]);
            sc.omega = (double)om;
            T rn = -om;
            rn = rn * (T)red[3];
            sc.rho_next = (double)rn;
          }
        }
      }
      __syncthreads();
      if (sc.err) break;
      // x += alpha p' (+ s omega) ; r = s - omega t ; |r|^2 ; BC fill (k_bicg_x, linalg.py:236-262)
      {
        const T omega = (T)sc.omega;
        const int early = sc.finished_early;
        v1[0] = 0.0;
        for (int c = tid; c < nbox; c += NT) {
          const unsigned m = meta[c];
          const int hx = RES_M_H(m), px = RES_M_P(m);
          T ap = alpha * H2p[hx];
          T xn = P1p[px] + ap;
          if (!early) {
            const T sv = Hp[hx];
            T so = sv * omega;
            xn = xn + so;
            T ot = omega * P3p[px];
            T rn = sv - ot;
            Hp[hx] = rn;
            T q = rn * rn;
            v1[0] += (double)q;
          }
          P1p[px] = xn;
        }
      }
      __syncthreads();
      if (!A.bc_static) {
        double unused = 0.0;
        bc_fill_and_shell(unused);
      }
      publish(Hp);
      if (!res_allreduce(S, A.parts, v1, red)) { timed_out = true; break; }
      if (tid == 0) {   // k_bicg_post stage 3
        if (sc.finished_early) {
          sc.done = 1;
        } else {
          T tol = (T)sqrt(red[0]);
          sc.tol = (double)tol;
          if (isnan(tol) || isinf(tol)) {
            sc.err = 1;
            sc.done = 1;
          } else {
            if (sc.tol <= sc.tolerance) sc.done = 1;
            if (sc.itr >= sc.max_it) sc.done = 1;
            T b = (T)sc.rho_next / (T)sc.rho;
            b = b * (T)sc.alpha;
            b = b / (T)sc.omega;
            sc.beta = (double)b;
            sc.rho = sc.rho_next;
          }
        }
      }
      __syncthreads();
      if (sc.done) break;
      receive(Hp);
      __syncthreads();
    } else {
      // ---- Jacobi sweep: x' = x + omega (b - A x) / diag(A) on S (k_jacobi) -------------------------
      double v[2] = {0.0, 0.0};
      for (int c = tid; c < nbox; c += NT) {
        const unsigned m = meta[c];
        const T xo = Hp[RES_M_H(m)];
        T xn = xo;
        if (RES_M_S(m)) {
          T diag = (T)0;
          const int nterms = LEAN ? 1 : E.nterms;   // LEAN: one term (a constant index keeps E off scratch memory)
          for (int q = 0; q < nterms; ++q) {
            const DevTerm<T>& t = E.t[LEAN ? 0 : q];
            T dg = (T)0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              if (!G.act[a]) continue;
              const int rc = RES_M_RC(m, a);
              T cB = E.lap.c23[a];
              if (!LEAN && E.rz && a == PA_RZ_AXIS) {
                int bi, bj, bk;
                decode(c, bi, bj, bk);
                cB = E.rz[2 * E.rz_n + LO(1) + bj];
              }
              if (rz_on && a == PA_RZ_AXIS) {
                int bi, bj, bk;
                decode(c, bi, bj, bk);
                cB = rzl[2 * rzs + bj];
              }
              T cC = (rc == 1 || rc == 2) ? -cB : E.lap.m2inv[a];
              dg = dg + cC;
            }
            if (t.has_coeff) {
              T cfv = t.coeff;
              if (!LEAN && t.coeff_f) {   // tensor coefficient Gamma(x) (fdm.py:166-169): read where it lives, device memory
                int bi, bj, bk;
                decode(c, bi, bj, bk);
                cfv = t.coeff_f[(int64_t)(LO(0) + bi) * G.s0 + (int64_t)(LO(1) + bj) * G.s1 + (LO(2) + bk)];
              }
              dg = dg * cfv;
            }
            dg = dg * t.sign;
            diag = diag + dg;
          }
          T ax = stencil(H, c, m, xo);
          T res = P2p[RES_M_P(m)] - ax;
          res = res / diag;
          T w = A.omega * res;
          xn = xo + w;
          if (!RES_M_SHELL(m)) {
            T df = xn - xo;
            T p2 = df * df;
            v[1] += (double)p2;
          }
        }
        P1p[RES_M_P(m)] = xn;
      }
      __syncthreads();
      for (int c = tid; c < nbox; c += NT) {
        const unsigned m = meta[c];
        Hp[RES_M_H(m)] = P1p[RES_M_P(m)];
      }
      __syncthreads();
      if (!A.bc_static) bc_fill_and_shell(v[1]);
      __syncthreads();
      publish(Hp);
      if (!res_allreduce(S, A.parts, v, red)) { timed_out = true; break; }
      if (tid == 0) pa_logic_jacobi<T>(&sc, red[1]);
      __syncthreads();
      if (sc.done) break;
      receive(Hp);
      __syncthreads();
    }
  }

  // ---- store ---------------------------------------------------------------------------------------
  // a grid-wide wait that timed out (workgroups descheduled for seconds ...): nothing is stored -- x, r and the
  // scalars in device memory are what they were before the launch, the `fail` flag tells the host to run the
  // launch-per-phase loop from there
#ifdef PA_RES_TIMING
  if (SOLVER == 0 && blockIdx.x == 0 && tid == 0 && tn > 0)
    printf("k_resident CG timing, ns per iteration over %lld iterations (100 MHz clock): passA %.0f step1 %.0f alpha %.0f passB %.0f "
           "bc %.0f publish %.0f step2 %.0f beta %.0f dnew %.0f\n", tn, 10.0 * tt[0] / tn, 10.0 * tt[1] / tn, 10.0 * tt[2] / tn,
           10.0 * tt[3] / tn, 10.0 * tt[4] / tn, 10.0 * tt[5] / tn, 10.0 * tt[6] / tn, 10.0 * tt[7] / tn, 10.0 * tt[8] / tn);
#endif
  if (timed_out) return;
  {   // ... also when another workgroup gave up in the very step that ended the solve here
    __shared__ int any_fail;
    if (tid == 0) any_fail = __hip_atomic_load(A.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (any_fail) return;
  }
  __syncthreads();
  for (int c = tid; c < nbox; c += NT) {
    int bi, bj, bk;
    decode(c, bi, bj, bk);
    const int i = LO(0) + bi, j = LO(1) + bj, k = LO(2) + bk;
    A.x[(int64_t)i * G.s0 + (int64_t)j * G.s1 + k] = X(i, j, k);
  }
  if (blockIdx.x == 0 && tid == 0) {
    sc.done = 1;
    *A.sc = sc;
  }
}

// ---- host side ------------------------------------------------------------------------------------------
struct ResPlan {
  int P[3], bmax[3], nface, G;
  unsigned o_h, o_p1, o_p2, o_bcc, o_sh, o_meta, o_lists, o_h2, o_h3, o_p3, o_rz, lds;
  int cells;
};

// tuning knobs of the plan (options "res_cells", "res_nt", "res_nt_cells": tests force other layouts; the defaults are
// what DESIGN.md quotes)
static int res_tune(int opt, int dflt) { return opt > 0 ? opt : dflt; }

static bool res_plan(const pa_ctx* c, size_t es, int solver, ResPlan& R) {
  const DevGeom& G = c->G;
  const int64_t N[3] = {G.n0, G.n1, G.n2};
  const int maxbox = RES_MAXBOX;
  const int per_wg = res_tune(c->res_cells, 1024);   // cells per workgroup aimed at (<= 64 workgroups)
  const int maxg = RES_MAXG;
  if (G.ncell > (int64_t)maxg * maxbox) return false;
  int P[3] = {1, 1, 1};
  const int want = (int)std::max<int64_t>(1, std::min<int64_t>(64, (G.ncell + per_wg - 1) / per_wg));
  for (;;) {
    int b[3];
    for (int a = 0; a < 3; ++a) b[a] = (int)((N[a] + P[a] - 1) / P[a]);
    const int g = P[0] * P[1] * P[2];
    const int cells = b[0] * b[1] * b[2];
    const int h[3] = {G.act[0] ? 1 : 0, G.act[1] ? 1 : 0, G.act[2] ? 1 : 0};
    const size_t halo = (size_t)(b[0] + 2 * h[0]) * (b[1] + 2 * h[1]) * (b[2] + 2 * h[2]);
    int nface = 1;   // largest box face among the faces normal to a mesh axis
    for (int a = 0; a < 3; ++a)
      if (G.act[a]) nface = std::max(nface, b[0] * b[1] * b[2] / b[a]);
    auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
    size_t o = 0;
    R.o_h = (unsigned)o;   o += up(halo * es);
    R.o_p1 = (unsigned)o;  o += up((size_t)cells * es);
    R.o_p2 = (unsigned)o;  o += up((size_t)cells * es);
    R.o_h2 = R.o_h3 = R.o_p3 = 0;
    if (solver == 2) {   // BiCGSTAB: r / s, p and v with halo ; x, r0, t without
      R.o_h2 = (unsigned)o; o += up(halo * es);
      R.o_h3 = (unsigned)o; o += up(halo * es);
      R.o_p3 = (unsigned)o; o += up((size_t)cells * es);
    }
    R.o_bcc = (unsigned)o; o += up((size_t)6 * nface * es);
    R.o_sh = (unsigned)o;  o += up((size_t)6 * nface * es);
    R.o_meta = (unsigned)o; o += up((size_t)cells * sizeof(unsigned));
    R.o_lists = (unsigned)o; o += up((size_t)6 * nface * (5 * sizeof(unsigned short) + sizeof(unsigned)));
    R.o_rz = (unsigned)o;
    if (c->coord != PA_COORD_XYZ) o += up((size_t)3 * b[1] * es);   // LEAN on an rz mesh: coefficient rows of the box
    // (the per-cell word has 13 bits for the haloed index, 12 for the plain one)
    const bool fits = cells <= maxbox && halo <= 8192 && o <= RES_LDS_LIMIT;
    auto accept = [&]() {
      for (int a = 0; a < 3; ++a) { R.P[a] = P[a]; R.bmax[a] = b[a]; }
      R.nface = nface; R.G = g; R.lds = (unsigned)o; R.cells = cells;
      return true;
    };
    if (fits && g >= want) return accept();
    // split the axis with the largest box extent; every box keeps >= 3 nodes per axis, so that the nodes a
    // face fill reads (prev, prev2) are in the box that holds the face node
    int best = -1;
    for (int a = 0; a < 3; ++a)
      if (G.act[a] && c->bc[2 * a].type != PA_BC_PERIODIC && N[a] / (P[a] + 1) >= 3 && (best < 0 || b[a] > b[best])) best = a;
    if (best < 0) return fits ? accept() : false;   // nothing left to cut (periodic / short axes): fewer, larger boxes
    P[best] += 1;
    if (P[0] * P[1] * P[2] > maxg) return false;
  }
}

static bool res_applicable(const pa_ctx* c) {
  if (!c->resident || c->slab || c->profile) return false;
  const DevGeom& G = c->G;
  for (int a = 0; a < 3; ++a) {   // every face has a BC; a periodic axis is periodic at both ends
    if (!G.act[a]) continue;
    const int lo = c->bc[2 * a].type, hi = c->bc[2 * a + 1].type;
    if (lo == PA_BC_NONE || hi == PA_BC_NONE) return false;
    if ((lo == PA_BC_PERIODIC) != (hi == PA_BC_PERIODIC)) return false;
  }
  // (tensor coefficients / speed fields, round 4: the general-equation build evaluates pa_apply_terms with the mesh's
  // node indices, so Gamma(x) and u(x) are read where they live -- device memory, L2 hits on meshes of this size -- while
  // the solver's own fields stay in LDS; central Div with a speed field reads u's wrap-around neighbours the same way)
  for (int a = 0; a < 3; ++a)
    if (G.act[a] && (a == 0 ? G.g0 : (a == 1 ? G.n1 : G.n2)) < 5) return false;
  return true;
}

template <typename T>
int pa_resident_launch(pa_ctx* c, int solver, T* x, const T* rhs, double tol, int64_t max_it, double omega) {
  PaRange range_("pyapes resident solve (one cooperative launch)");
  if (!res_applicable(c)) return 0;
  ResPlan R;
  if (!res_plan(c, sizeof(T), solver, R)) return 0;
  // LEAN: one Laplacian term with a scalar coefficient -- on an xyz mesh, or on an axisymmetric one (the r axis takes
  // its rows from the table of pa_coord_set, staged in LDS)
  // (option "res_rzlean" 0: pa_apply_terms on the box, as before round 3 -- tests)
  const bool rz_lean = c->coord != PA_COORD_XYZ && c->rz_tab && c->res_rzlean;
  const bool lean = c->nterms == 1 && c->terms[0].kind == PA_OP_LAPLACIAN && !c->terms[0].coeff_field &&
                    (c->coord == PA_COORD_XYZ || rz_lean);
  // threads per workgroup: more waves hide the LDS latency of the cell passes once a thread has several cells;
  // the general-equation build needs more registers than 512 / 1024 threads leave
  int nt = 256;
  if (lean && R.cells > res_tune(c->res_nt_cells, 256))
    nt = res_tune(c->res_nt, R.cells > 1024 ? 1024 : 512);
  const void* fn;
  auto pick = [&](const void* f0, const void* f1, const void* f2) { return solver == 0 ? f0 : (solver == 1 ? f1 : f2); };
  if (!lean) fn = pick((const void*)k_resident<T, 0, false, 256>, (const void*)k_resident<T, 1, false, 256>, (const void*)k_resident<T, 2, false, 256>);
  else if (nt == 1024) fn = pick((const void*)k_resident<T, 0, true, 1024>, (const void*)k_resident<T, 1, true, 1024>, (const void*)k_resident<T, 2, true, 1024>);
  else if (nt == 512) fn = pick((const void*)k_resident<T, 0, true, 512>, (const void*)k_resident<T, 1, true, 512>, (const void*)k_resident<T, 2, true, 512>);
  else { nt = 256; fn = pick((const void*)k_resident<T, 0, true, 256>, (const void*)k_resident<T, 1, true, 256>, (const void*)k_resident<T, 2, true, 256>); }
  // co-residency: one workgroup per CU at this LDS size; the cooperative launch itself refuses a grid that does
  // not fit (then the launch-per-phase loops run).  The attribute / occupancy queries are made once per kernel and
  // LDS size (they cost more host time than the launch).
  struct Fit { const void* fn; unsigned lds; int nt, device, max_grid; };
  static Fit fits[64];
  static int nfits = 0;
  static std::mutex fits_mu;   // ctxs of different threads share the table
  std::lock_guard<std::mutex> lock(fits_mu);
  int max_grid = -1;
  for (int q = 0; q < nfits; ++q)
    if (fits[q].fn == fn && fits[q].lds == R.lds && fits[q].nt == nt && fits[q].device == c->device) max_grid = fits[q].max_grid;
  if (max_grid < 0) {
    max_grid = 0;
    int per_cu = 0, cus = 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RES_LDS_LIMIT) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, nt, R.lds) == hipSuccess && per_cu >= 1 &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess)
      max_grid = per_cu * cus;
    else
      (void)hipGetLastError();
    if (nfits < 64) fits[nfits++] = Fit{fn, R.lds, nt, c->device, max_grid};
  }
  if (max_grid < R.G) return 0;
  // scratch: counter + fail flag | partial sums | mailboxes
  const size_t head = 256;
  const size_t parts_b = (size_t)2 * R.G * RES_NS * sizeof(double);
  const size_t mail_b = (size_t)2 * R.G * 6 * R.nface * sizeof(T);
  int rc;
  if ((rc = pa_scratch(c, &c->scr[SCR_RES], &c->cap[SCR_RES], head + parts_b + mail_b))) return rc;
  char* base = (char*)c->scr[SCR_RES];
  PA_HIP(c, hipMemsetAsync(base, 0, head + parts_b, c->stream));

  ResArgs<T> A;
  memset(&A, 0, sizeof(A));
  for (int w = 0; w < 6; ++w) A.f[w].face = -1;
  int nw = 0;
  for (int w = 0; w < c->nbc && nw < 6; ++w) {
    const int f = c->bc_order[w];
    const HostBC& b = c->bc[f];
    if (b.type == PA_BC_NONE || !c->G.act[f >> 1]) continue;
    ResFace<T>& F = A.f[nw++];
    F.face = f;
    F.type = b.type;
    F.vals = (const T*)b.vals;
    F.dxf = (T)b.dxf;
    F.ndir = (f & 1) == 0 ? (T)-1 : (T)1;
    if (b.type == PA_BC_DIRICHLET) F.sval = (T)b.value;
    if (b.type == PA_BC_NEUMANN) {   // as pa_bc_apply_faces
      T pre = (T)((2.0 / 3.0) * b.value);
      pre = pre * F.dxf;
      pre = pre * F.ndir;
      F.sval = pre;
    }
  }
  A.c43 = (T)(4.0 / 3.0);
  A.c13 = (T)(1.0 / 3.0);
  A.c23 = (T)(2.0 / 3.0);
  for (int a = 0; a < 3; ++a) { A.P[a] = R.P[a]; A.bmax[a] = R.bmax[a]; }
  A.nface = R.nface;
  A.bc_static = pa_bc_is_static(c) ? 1 : 0;
  A.omega = (T)omega;
  A.x = x; A.rhs = rhs;
  A.tol = tol; A.max_it = (long long)max_it;
  A.x_old_out = (T*)c->x_old_out;
  A.sc = c->sc;
  A.counter = (unsigned long long*)base;
  A.fail = (int*)(base + 64);
  A.parts = (double*)(base + head);
  A.mail = (T*)(base + head + parts_b);
  A.spin_max = 1u << 21;   // x (one atomic load + s_sleep) ~ seconds: far beyond any legitimate wait
  if (c->res_spin >= 0) A.spin_max = (unsigned)c->res_spin;   // option "res_spin" (tests: 0 = every wait gives up)
  A.o_h = R.o_h; A.o_p1 = R.o_p1; A.o_p2 = R.o_p2; A.o_bcc = R.o_bcc; A.o_sh = R.o_sh; A.o_meta = R.o_meta; A.o_lists = R.o_lists;
  A.o_h2 = R.o_h2; A.o_h3 = R.o_h3; A.o_p3 = R.o_p3;
  A.o_rz = R.o_rz;
  A.rz_on = (lean && c->coord != PA_COORD_XYZ) ? 1 : 0;
  DevGeom G = c->G;
  DevEq<T> E;
  pa_build_eq<T>(c, c->nterms, c->terms, E);
  void* args[] = {&G, &E, &A};
  hipError_t e = c->resident_coop ? hipLaunchCooperativeKernel(fn, dim3(R.G), dim3(nt), args, R.lds, c->stream)
                                  : hipLaunchKernel(fn, dim3(R.G), dim3(nt), args, R.lds, c->stream);
  if (e != hipSuccess) {   // e.g. the device is shared and the grid cannot be co-resident right now
    (void)hipGetLastError();
    return 0;
  }
  return R.G;
}

extern "C" {
int pa_resident_used(const pa_ctx* c) { return c ? c->resident_used : 0; }

int pa_resident_plan(pa_ctx* c, int solver, int* boxes) {   // what a solve on the bound mesh would use (tests, DESIGN numbers)
  if (!c || !boxes || solver < 0 || solver > 2) return PA_E_ARG;
  if (!c->grid_set || !c->eq_set) { pa_set_err(c, "pa_resident_plan: grid / equation not set"); return PA_E_STATE; }
  ResPlan R;
  boxes[0] = boxes[1] = boxes[2] = 0;
  if (!res_applicable(c) || !res_plan(c, (size_t)c->esize, solver, R)) return 0;
  for (int a = 0; a < 3; ++a) boxes[a] = R.P[a];
  return R.G;
}
}  // extern "C"

template int pa_resident_launch<float>(pa_ctx*, int, float*, const float*, double, int64_t, double);
template int pa_resident_launch<double>(pa_ctx*, int, double*, const double*, double, int64_t, double);
