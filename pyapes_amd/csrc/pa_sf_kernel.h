// pa_sf_kernel.h -- k_sf, the marching kernel of the instruction-heavy SINGLE-FIELD operations: everything
// with a Div term -- the explicit Euler step, Div, Laplacian + Div.  These move 8-24 bytes per cell, a third of
// a CG phase, and carry 2-3 times its arithmetic (no FMA: every product and sum is rounded separately), so
// what bounds them is instruction issue and how well it overlaps the loads.  k_cg3d (pa_cg3d_kernel.h) stages
// every plane in LDS behind a workgroup barrier and needs ~250 VGPRs at four rows per thread -- two waves per
// SIMD advancing in lock step; its Euler step ran at 0.49 of the HBM roofline whatever the tile height or the
// number of workgroups per CU, 46 VALU instructions per cell.
//
// Here every WAVE marches on its own: no LDS, no barrier.
//   * A wave owns RJ rows x 64 lanes x VEC cells (VEC = 16 B / sizeof(T): one 16-byte lane access per row,
//     1 KiB per wave and row) and keeps four planes of them in registers -- behind / current / ahead / the
//     one being loaded -- addressed by compile-time slot numbers (the plane loop is unrolled by four, so
//     nothing is ever rotated).
//   * j +- 1 across the wave's row block: the two rows above / below are loaded by the wave itself.  They
//     are the neighbour wave's own rows, at the same time on the same XCD (blockIdx -> tile is XCD-aware):
//     L2 hits, 8.2 B of HBM traffic per cell measured against 8 algorithmic.
//   * k +- 1 across lanes: one DPP move per row and side (v_mov_b32 wave_shr:1 / wave_shl:1); lane 0 / 63
//     keep the cell left / right of the tile, which every lane loads with one (uniform-address) access.
//   * loads run two planes ahead of the arithmetic, and the code between their issue and their use is free
//     of branches (see `plane`): that is what lets the compiler wait with vmcnt(N > 0).
// Result (MI355X): explicit Euler step 512^3 fp32 273 -> 209 us (0.64 of the 8 TB/s roofline; a plain device
// copy of the array: 0.66), 256^3 (BASELINE config 4) 38.7 -> 28.3 us; upwind Div fp64 512^3 553 -> 428 us.
// Arithmetic: the row expressions of k_cg3d's VROW path, operation for operation (-ffp-contract=off), so
// results are bit-identical to it, to the generic kernels and to the oracle (tests/test_gpu_tiled_ops.py,
// test_gpu_parity_golden.py, test_gpu_properties.py).
#pragma once
#include "pa_cg3d_kernel.h"

#include <type_traits>

template <typename T> struct SfBits;
template <> struct SfBits<float> {
  static __device__ __forceinline__ float prev(float x, float edge) {   // value of lane - 1; lane 0 keeps `edge`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(x), 0x138, 0xf, 0xf, false));
  }
  static __device__ __forceinline__ float next(float x, float edge) {   // value of lane + 1; lane 63 keeps `edge`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(x), 0x130, 0xf, 0xf, false));
  }
};
template <> struct SfBits<double> {
  static __device__ __forceinline__ double mv(double x, double edge, bool up) {
    const long long xb = __double_as_longlong(x), eb = __double_as_longlong(edge);
    int lo, hi;
    if (up) {
      lo = __builtin_amdgcn_update_dpp((int)eb, (int)xb, 0x138, 0xf, 0xf, false);
      hi = __builtin_amdgcn_update_dpp((int)(eb >> 32), (int)(xb >> 32), 0x138, 0xf, 0xf, false);
    } else {
      lo = __builtin_amdgcn_update_dpp((int)eb, (int)xb, 0x130, 0xf, 0xf, false);
      hi = __builtin_amdgcn_update_dpp((int)(eb >> 32), (int)(xb >> 32), 0x130, 0xf, 0xf, false);
    }
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
  }
  static __device__ __forceinline__ double prev(double x, double edge) { return mv(x, edge, true); }
  static __device__ __forceinline__ double next(double x, double edge) { return mv(x, edge, false); }
};

// (Round 4, tried and removed: packed fp32 subtraction.  There is no v_pk_sub_f32, and the compiler issues four v_sub_f32
// for a row of four cells -- 28 of the 104 VALU instructions of an Euler row.  a + (-b) through v_pk_add_f32 with the
// negation as an operand modifier (inline asm) took the body from ~120 to ~98 VALU instructions, and the kernel from 158 /
// 159 to 172 / 174 VGPRs, i.e. from three waves per SIMD to two: 256^3 fp32 31.4 against 31.6 us per step, c4t 35.9 / 35.9,
// 512^3 210.0 / 210.4 -- nothing.  Forcing three waves (amdgpu_waves_per_eu) spills to scratch memory.  At 512^3 the step
// runs at the speed of a plain copy of its arrays (210 us against 205); at 256^3, where both arrays sit in the Infinity
// Cache and a copy takes 19.5 us, it is bound by issue AND by the waves it can keep in flight.  Tried again on top of the US
// kernels below (153 -> 164 VGPRs, still three waves per SIMD): 27.96 / 28.10, 28.08 / 28.19, 29.22 / 28.15 us -- with the
// dead upwind half gone the step no longer responds to instruction counts.)
// PHASE 2: A x (KIND 0: Laplacian; else Laplacian + Div, or the Div term alone when lap_off)
// PHASE 3: explicit Euler step (KIND = PA_OP_DIV_* of the advection term)
// PHASE 7: explicit gradient
// HASU: the advection speed is a field (read at the cell)
//
// Instruction economy (a wave64 VALU instruction occupies its SIMD for four cycles whatever it does): in-plane
// positions are 32-bit byte offsets added to a uniform plane pointer, everything that does not change from
// plane to plane -- row / column coefficients, masks, u+ / u- of a scalar speed -- is formed once in front of
// the loop, planes beside an axis-0 face take a second copy of the body so that the common one has no
// coefficient selects, and there is no interior-set select: k_sf only takes launches whose output is wanted
// at every node (sf_applies).
//
// BCL ("BC on load", the explicit Euler MARCH without a BC-fill launch per step).  The stencil at a node of the
// interior set reads boundary nodes only at face INTERIORS -- a node on an edge or corner is never the neighbour of
// an interior node -- and what the ordered BC fill stores there is a function of the two nodes behind it along the
// face normal (bcs.py:200-262: dirichlet g; neumann 4/3 x[prev] - 1/3 x[prev2] + (2/3) V dx n; symmetry x[prev]).
// For a node with index 1 (or N - 2) on an axis those two nodes are the node itself and its inner neighbour --
// operands the stencil holds anyway.  So a step of the march does not need the boundary values of its input at all:
// it REPLACES the outer operand by the face formula (same operations, same order, same bits as the fill kernels
// of pa_bc.hip), the boundary nodes of the intermediate states stay unfilled, and ONE ordered fill after the last
// step makes the result's boundary what the step-by-step sequence leaves there.  Without periodic faces only
// (their fill reads the far end of the axis and puts boundary nodes into the interior set).
// US (round 4): sign of a SCALAR advection speed of the upwind scheme, known at launch -- 1: u >= 0, 2: u < 0, 0: not used
// (speed field, other schemes).  One of u+ = max(u, 0) / u- = min(u, 0) is then zero and its half of every axis term
//   t = u+ (x - x[-1]) + u- (x[+1] - x)
// is a signed zero: t = u+ (x - x[-1]) + (+-0) has the value of its first product, and if that is itself +-0 only the sign
// of a zero can differ -- which the accumulation adv = (+0) + t_0 + t_1 + t_2 absorbs (+0 + -0 = +0, and a sum that starts
// at +0 never becomes -0).  So the dead half -- one subtraction, one product, one addition per axis, 9 of the ~42 row
// operations of an Euler step -- is not computed, and every FINITE field gives the same bits (tests/test_gpu_tiled_ops.py
// against the generic kernels, which form both halves).  Only a non-finite neighbour on the dead side differs: 0 x inf is
// NaN in the literal form, nothing here.
#ifndef PA_SF_USIGN
#define PA_SF_USIGN 1
#endif
template <typename T, int RJ, int PHASE, int KIND, bool HASU, bool BCL = false, int US = 0>
__global__ void __launch_bounds__(256) k_sf(Cg3dArgs<T> A) {
  static_assert(!BCL || PHASE == 3, "BC on load: the Euler step");
  static_assert(US == 0 || (KIND == 4 && !HASU), "US: scalar speed of the upwind scheme");
  constexpr int VEC = VecOf<T>::N;
  typedef T V __attribute__((ext_vector_type(VEC)));
  constexpr int TJ = 4 * RJ, TK = 64 * VEC;
  static_assert(PHASE == 2 || PHASE == 3 || PHASE == 7, "single-field phases");   // 7 and KIND 0 are not
  // instantiated: the gradient and the Laplacian alone run at copy speed on k_cg3d (pa_sf.hip)
  constexpr bool DIV = (PHASE == 3) || (PHASE == 2 && KIND != 0);
  const DevGeom& G = A.G;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int vb = pa_xcd_remap(blockIdx.x, gridDim.x);
  const int tiles = A.tiles_j * A.tiles_k;
  const int chunk = vb / tiles, tl = vb - chunk * tiles;
  const int tjb = tl / A.tiles_k, tkb = tl - tjb * A.tiles_k;
  const int n0 = (int)G.n0, n1 = (int)G.n1, n2 = (int)G.n2;
  const int i0 = (int)((int64_t)chunk * n0 / A.chunks), i1 = (int)((int64_t)(chunk + 1) * n0 / A.chunks);
  const int CI = i1 - i0;
  const int j0 = tjb * TJ + wv * RJ, k0 = tkb * TK;
  auto wrap = [](int v, int n) { v %= n; return v < 0 ? v + n : v; };

  // ---- per-thread geometry: RJ rows x VEC columns, as 32-bit byte offsets inside a plane ------------
  const int kg = k0 + lane * VEC;
  const int kc = wrap(kg, n2);
  const bool kvalid = kg < n2;
  unsigned off[RJ], offe[RJ];
  const int ecol = lane == 63 ? wrap(k0 + TK, n2) : wrap(k0 - 1, n2);   // lane 0: cell left of the tile, 63: right
  bool rowOk[RJ];
  T cPj[RJ], cCj[RJ], cMj[RJ];     // Laplacian rows along j (uniform per row)
  T gPj[RJ], gCj[RJ], gMj[RJ];     // gradient rows along j
  bool rPLo[RJ], rPHi[RJ];
  // BCL, by what the register allocator makes of it (measured, us per step, classic -> BCL).  Faces normal to the
  // march axis and to j are whole ROWS of a wave: their fill values replace the loaded rows in registers, under a
  // wave-uniform branch, and the stencil body is untouched (PATCH_IJ, both precisions).  The two k-face cells of a row:
  //   PATCH_K (fp64)  replaced in the row's registers as well: VGPRs as the plain kernel (156 / 236); 256^3 66 -> 54.
  //   SUBST_K (fp32)  single components of packed rows do not patch cheaply (+40 VGPRs, slower than the classic
  //                   sequence: 256^3 35 -> 41) -- the stencil's k - 1 / k + 1 operand of cells 1 / n2 - 2 is replaced
  //                   where it is read instead (one select per row when no k face is a Neumann face).
  //   (round 3, first form: every face substituted at the operand, fp32 -- 171 / 248 VGPRs, 706 v_readlane of spilled
  //   SGPRs, 6,100 instructions against 2,800 for the plain kernel; 256^3 34.8 -> 33.3)
  constexpr bool PATCH = BCL, PATCH_K = BCL && sizeof(T) == 8, SUBST = BCL && !PATCH_K;
  bool rTop[RJ];                    // PATCH: the row is the upper j face (its value is formed from the two rows below)
#pragma unroll
  for (int jj = 0; jj < RJ; ++jj) {
    const int jg = j0 + jj;
    rTop[jj] = PATCH && jg == n1 - 1 && A.bcl_type[3] != 0;
    const unsigned ro = (unsigned)wrap(jg, n1) * (unsigned)n2;
    off[jj] = (ro + (unsigned)kc) * (unsigned)sizeof(T);
    offe[jj] = (ro + (unsigned)ecol) * (unsigned)sizeof(T);
    rowOk[jj] = kvalid && jg < n1;
    const int rc = pa_row_case(G, 1, jg, G.n1, G.treat);
    cPj[jj] = A.lap.inv[1]; cCj[jj] = A.lap.m2inv[1]; cMj[jj] = A.lap.inv[1];
    if (rc == 1) { cPj[jj] = A.lap.c23[1]; cCj[jj] = -A.lap.c23[1]; cMj[jj] = (T)0; }
    if (rc == 2) { cPj[jj] = (T)0; cCj[jj] = -A.lap.c23[1]; cMj[jj] = A.lap.c23[1]; }
    rPLo[jj] = G.bct[2] == 4 && jg == 1;
    rPHi[jj] = G.bct[3] == 4 && jg == n1 - 2;
    if (PHASE == 7) {
      gPj[jj] = A.grd.g[1]; gCj[jj] = (T)0; gMj[jj] = A.grd.mg[1];
      if (rc == 1) { gPj[jj] = A.grd.lo_p[1]; gCj[jj] = A.grd.lo_c[1]; gMj[jj] = (T)0; }
      if (rc == 2) { gPj[jj] = (T)0; gCj[jj] = A.grd.hi_c[1]; gMj[jj] = A.grd.hi_m[1]; }
      if (rPLo[jj]) gMj[jj] = (T)0;
      if (rPHi[jj]) gPj[jj] = (T)0;
    }
  }
  const unsigned offu = ((unsigned)wrap(j0 - 1, n1) * (unsigned)n2 + (unsigned)kc) * (unsigned)sizeof(T);
  const unsigned offd = ((unsigned)wrap(j0 + RJ, n1) * (unsigned)n2 + (unsigned)kc) * (unsigned)sizeof(T);
  bool cPLo[VEC], cPHi[VEC];
  V cPkV, cCkV, cMkV, gPkV, gCkV, gMkV;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const int kk = kg + v;
    const int rc = pa_row_case(G, 2, kk, G.n2, G.treat);
    T p = A.lap.inv[2], c0 = A.lap.m2inv[2], mq = A.lap.inv[2];
    if (rc == 1) { p = A.lap.c23[2]; c0 = -A.lap.c23[2]; mq = (T)0; }
    if (rc == 2) { p = (T)0; c0 = -A.lap.c23[2]; mq = A.lap.c23[2]; }
    cPkV[v] = p; cCkV[v] = c0; cMkV[v] = mq;
    cPLo[v] = G.bct[4] == 4 && kk == 1;
    cPHi[v] = G.bct[5] == 4 && kk == n2 - 2;
    if (PHASE == 7) {
      T gp = A.grd.g[2], gc = (T)0, gm = A.grd.mg[2];
      if (rc == 1) { gp = A.grd.lo_p[2]; gc = A.grd.lo_c[2]; gm = (T)0; }
      if (rc == 2) { gp = (T)0; gc = A.grd.hi_c[2]; gm = A.grd.hi_m[2]; }
      if (cPLo[v]) gm = (T)0;
      if (cPHi[v]) gp = (T)0;
      gPkV[v] = gp; gCkV[v] = gc; gMkV[v] = gm;
    }
  }
  // BCL.  The boundary VALUES of the input are replaced in registers, once per plane, when the plane becomes the
  // current one (its boundary rows / cells feed the j / k neighbours of its interior nodes) and, for the two axis-0
  // faces, when the plane next to the face is current (the face plane is its i - 1 / i + 1 operand); the stencil
  // body is untouched.  Along k: cell 0 is component 0 of the lane that starts the row, cell n2 - 1 the last
  // component of the lane that ends it (rows are whole vectors: sf_applies).  (Forming the k-face cells on STORE
  // instead -- from the row's new values -- was tried: 282 / 320 VGPRs for the fp32 kernels against 199 / 268.)
  const bool cBLo = BCL && kg == 0 && A.bcl_type[4] != 0;
  const bool cBHi = BCL && kg + VEC == n2 && A.bcl_type[5] != 0;
  const bool jBot = PATCH && j0 == 0 && A.bcl_type[2] != 0;
  // the value the ordered fill stores on face f, from the node behind it (a) and the one behind that (b): the
  // operations of k_bc_face / k_bc_compute in their order.  The face's type is uniform: no arithmetic for the
  // faces that need none.
  // (whole rows under a wave-uniform branch -- the planes / rows on a face are few; along k, where every row has
  // its two cells, the arithmetic of a neumann face sits behind ONE uniform branch and the rest is selects.
  // First attempt: the substitution inside the stencil body, per operand -- 170-280 VGPRs instead of 157-248, a
  // wave less per SIMD, the step 29 % slower)
  auto bcvV = [&](int f, const V& a, const V& b) -> V {
    const int ty = A.bcl_type[f];
    if (ty == 2) {
      V t1 = A.bcl_c43 * a;
      V t2 = A.bcl_c13 * b;
      t1 = t1 - t2;
      return t1 + A.bcl_val[f];
    }
    if (ty == 1) return (V)A.bcl_val[f];
    return a;
  };
  const bool kNeu = BCL && (A.bcl_type[4] == 2 || A.bcl_type[5] == 2);
  const size_t pstride = (size_t)G.s0 * sizeof(T);
  // Uniform and BRANCH-FREE: with a branch between the issue of a load and its use the compiler's waitcnt
  // pass loses count of what is outstanding and falls back to s_waitcnt vmcnt(0) right behind the loads --
  // the two-plane software pipeline is then gone without a word (seen: 240 -> 279 us at 512^3 fp32).  For
  // the same reason every load below is unconditional, on a clamped plane index.
  // (integer selects lose the address space: the result is cast back to a GLOBAL pointer explicitly, or the
  // loads become flat_load, which also count on lgkmcnt)
  typedef const char __attribute__((address_space(1))) * gcptr;
  auto plane = [&](int ii) -> gcptr {
    const int ic = ii < 0 ? 0 : (ii >= n0 ? n0 - 1 : ii);
    uintptr_t u = (uintptr_t)A.d.p + (size_t)(unsigned)ic * pstride;
    const uintptr_t mlo = (uintptr_t)0 - (uintptr_t)(ii < 0), mhi = (uintptr_t)0 - (uintptr_t)(ii >= n0);
    u = (u & ~mlo) | ((uintptr_t)A.d.glo & mlo);
    u = (u & ~mhi) | ((uintptr_t)A.d.ghi & mhi);
    return (gcptr)u;
  };

  // scalar speed: u+ / u- (upwind) or the three rows (literal upwind) once
  V uplC = (V)(T)0, umiC = (V)(T)0;
  if (DIV && !HASU) {
    const T u = A.u;
    uplC = (V)(u > (T)0 ? u : (T)0);
    umiC = (V)(u < (T)0 ? u : (T)0);
  }

  // ---- register planes -----------------------------------------------------------------------------
  V P[4][RJ];          // own rows of four planes: slot of chunk-relative plane q is (q + 1) & 3
  V Hu[4], Hd[4];      // rows above / below: slot q & 3 (loaded two planes ahead, like the own rows)
  T He[4][RJ];         // edge cells of those planes
  V U[4][HASU ? RJ : 1];   // advection speed field of those planes

  auto load_own = [&](auto SLOT, int ii) {
    constexpr int s = decltype(SLOT)::value;
    gcptr p = plane(ii);
#pragma unroll
    for (int jj = 0; jj < RJ; ++jj) P[s][jj] = *reinterpret_cast<const V __attribute__((address_space(1)))*>(p + off[jj]);
  };
  auto load_halo = [&](auto SLOT, int ii) {
    constexpr int s = decltype(SLOT)::value;
    gcptr p = plane(ii);
    Hu[s] = *reinterpret_cast<const V __attribute__((address_space(1)))*>(p + offu);
    Hd[s] = *reinterpret_cast<const V __attribute__((address_space(1)))*>(p + offd);
#pragma unroll
    for (int jj = 0; jj < RJ; ++jj) He[s][jj] = *reinterpret_cast<const T __attribute__((address_space(1)))*>(p + offe[jj]);
    if constexpr (HASU) {
      const char* pu = (const char*)A.aux + (size_t)ii * pstride;
#pragma unroll
      for (int jj = 0; jj < RJ; ++jj) U[s][jj] = *reinterpret_cast<const V*>(pu + off[jj]);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  // prologue: planes -1, 0, 1 of the chunk and the halo of plane 0
  load_own(I0{}, i0 - 1);
  load_own(I1{}, i0);
  load_halo(I0{}, i0);
  load_own(I2{}, i0 + 1);   // also for a chunk of one plane: the next chunk's first plane / the ghost
  load_halo(I1{}, i0 + 1 < i1 ? i0 + 1 : i1 - 1);

  const T sgn = A.sign;
  const T cf = A.has_coeff ? A.coeff : (T)1;   // x * 1 is x: no select in the loop
  // k_sf writes the operator's value at EVERY node: the Euler step where the caller's BC fill rewrites all
  // nodes outside the interior set (out_all), A x without interior_only.  The masked forms stay on k_cg3d
  // (sf_applies), so the loop carries no interior-set select at all.
  const int gcomp0 = A.gnd - 3;

  // one plane: C = q & 3 (q = chunk-relative plane index); slots behind C, current C+1, ahead C+2, loading C+3
  auto step = [&](auto CC, int q) {
    constexpr int C = decltype(CC)::value;
    constexpr int SB = C & 3, SC = (C + 1) & 3, SA = (C + 2) & 3, SL = (C + 3) & 3, HC = C & 3, HN = (C + 2) & 3;
    const int ii = i0 + q;
    // loads for later planes first: they fly during this plane's arithmetic
    load_own(std::integral_constant<int, SL>{}, ii + 2 <= i1 ? ii + 2 : i1);          // plane q + 2 (<= the one behind the chunk)
    load_halo(std::integral_constant<int, HN>{}, ii + 2 < i1 ? ii + 2 : i1 - 1);
    // keep them HERE: left to itself the scheduler sinks these loads to just above their first use two planes
    // later (shorter live ranges, 163 -> 116 VGPRs) and the software pipeline is gone (512^3 fp32: 240 -> 279 us)
    __builtin_amdgcn_sched_barrier(0);

    // the planes next to an axis-0 face carry other coefficients (and the first / last plane may lie outside
    // the interior set): they take the same code with the values selected; every other plane takes it with
    // the plain constants, free of selects
    const int64_t gi = ii + G.off0;
    const int rci = pa_row_case(G, 0, gi, G.g0, G.treat);
    const bool iPLo_ = G.bct[0] == 4 && gi == 1;
    const bool iPHi_ = G.bct[1] == 4 && gi == G.g0 - 2;
    const bool iBLo_ = BCL && gi == 1 && A.bcl_type[0] != 0;
    const bool iBHi_ = BCL && gi == G.g0 - 2 && A.bcl_type[1] != 0;
    char* const po = (char*)A.out + (size_t)ii * pstride;
    if constexpr (PATCH) {
      if constexpr (PATCH_K) {
#pragma unroll
      for (int jj = 0; jj < RJ; ++jj) {   // k faces of the current plane's rows
        V& r = P[SC][jj];
        T lo = r[1], hi = r[VEC - 2];     // symmetry: x[face] = x[prev]
        if (kNeu) {
          const T in2 = VEC > 2 ? r[VEC > 2 ? 2 : 0] : SfBits<T>::next(r[0], r[0]);               // cell 2
          const T in3 = VEC > 2 ? r[VEC > 2 ? VEC - 3 : 0] : SfBits<T>::prev(r[VEC - 1], r[0]);   // cell n2 - 3
          T t1 = A.bcl_c43 * r[1];
          T t2 = A.bcl_c13 * in2;
          t1 = t1 - t2;
          t1 = t1 + A.bcl_val[4];
          lo = A.bcl_type[4] == 2 ? t1 : lo;
          t1 = A.bcl_c43 * r[VEC - 2];
          t2 = A.bcl_c13 * in3;
          t1 = t1 - t2;
          t1 = t1 + A.bcl_val[5];
          hi = A.bcl_type[5] == 2 ? t1 : hi;
        }
        lo = A.bcl_type[4] == 1 ? A.bcl_val[4] : lo;
        hi = A.bcl_type[5] == 1 ? A.bcl_val[5] : hi;
        r[0] = cBLo ? lo : r[0];
        r[VEC - 1] = cBHi ? hi : r[VEC - 1];
      }
      }
      if (jBot) P[SC][0] = bcvV(2, P[SC][RJ > 1 ? 1 : 0], RJ > 2 ? P[SC][RJ > 2 ? 2 : 0] : Hd[HC]);   // row 0 from rows 1, 2
#pragma unroll
      for (int jj = 1; jj < RJ; ++jj)      // row n1 - 1 from the two rows below it (same block: launch_sf_any)
        if (rTop[jj]) P[SC][jj] = bcvV(3, P[SC][jj - 1], jj >= 2 ? P[SC][jj >= 2 ? jj - 2 : 0] : Hu[HC]);
      if (iBLo_) {
#pragma unroll
        for (int jj = 0; jj < RJ; ++jj) P[SB][jj] = bcvV(0, P[SC][jj], P[SA][jj]);
      }
      if (iBHi_) {
#pragma unroll
        for (int jj = 0; jj < RJ; ++jj) P[SA][jj] = bcvV(1, P[SC][jj], P[SB][jj]);
      }
    }
    auto body = [&](auto PLAINC) {
    constexpr bool PLAIN = decltype(PLAINC)::value;
    const bool iPLo = PLAIN ? false : iPLo_, iPHi = PLAIN ? false : iPHi_;
    T cPi = A.lap.inv[0], cCi = A.lap.m2inv[0], cMi = A.lap.inv[0];
    T gP0 = A.grd.g[0], gC0 = (T)0, gM0 = A.grd.mg[0];
    if (!PLAIN) {
      if (rci == 1) { cPi = A.lap.c23[0]; cCi = -A.lap.c23[0]; cMi = (T)0; }
      if (rci == 2) { cPi = (T)0; cCi = -A.lap.c23[0]; cMi = A.lap.c23[0]; }
      if (PHASE == 7) {
        if (rci == 1) { gP0 = A.grd.lo_p[0]; gC0 = A.grd.lo_c[0]; gM0 = (T)0; }
        if (rci == 2) { gP0 = (T)0; gC0 = A.grd.hi_c[0]; gM0 = A.grd.hi_m[0]; }
        if (iPLo) gM0 = (T)0;
        if (iPHi) gP0 = (T)0;
      }
    }
#pragma unroll
    for (int jj = 0; jj < RJ; ++jj) {
      const V xc = P[SC][jj];
      V xpi = P[SA][jj], xmi = P[SB][jj];
      V up, dn;   // rows j-1 / j+1
      if (jj > 0) up = P[SC][jj - 1]; else up = Hu[HC];
      if (jj < RJ - 1) dn = P[SC][jj + 1]; else dn = Hd[HC];
      // k-1 / k+1: inside the lane's vector, across lanes by DPP (lane 0 / 63 keep the tile's edge cell)
      V xpk, xmk;
      const T edge = He[HC][jj];
      const T fromPrev = SfBits<T>::prev(xc[VEC - 1], edge);
      const T fromNext = SfBits<T>::next(xc[0], edge);
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        xpk[v] = (v < VEC - 1) ? xc[v + 1 < VEC ? v + 1 : v] : fromNext;
        xmk[v] = (v > 0) ? xc[v > 0 ? v - 1 : 0] : fromPrev;
      }
      if constexpr (SUBST) {   // cells 1 / n2 - 2 (one lane each): per-lane selects
        T lo = xc[1], hi = xc[VEC - 2];   // symmetry: x[face] = x[prev]
        if (kNeu) {
          T t1 = A.bcl_c43 * xc[1];
          T t2 = A.bcl_c13 * xpk[1];
          t1 = t1 - t2;
          t1 = t1 + A.bcl_val[4];
          lo = A.bcl_type[4] == 2 ? t1 : lo;
          t1 = A.bcl_c43 * xc[VEC - 2];
          t2 = A.bcl_c13 * xmk[VEC - 2];
          t1 = t1 - t2;
          t1 = t1 + A.bcl_val[5];
          hi = A.bcl_type[5] == 2 ? t1 : hi;
        }
        lo = A.bcl_type[4] == 1 ? A.bcl_val[4] : lo;
        hi = A.bcl_type[5] == 1 ? A.bcl_val[5] : hi;
        xmk[1] = cBLo ? lo : xmk[1];
        xpk[VEC - 2] = cBHi ? hi : xpk[VEC - 2];
      }
      if constexpr (PHASE == 7) {
        V s = gP0 * xpi;
        V mm = gC0 * xc;
        s = s + mm;
        mm = gM0 * xmi;
        const V g0 = s + mm;
        s = gPj[jj] * dn;
        mm = gCj[jj] * xc;
        s = s + mm;
        mm = gMj[jj] * up;
        const V g1 = s + mm;
        s = gPkV * xpk;
        mm = gCkV * xc;
        s = s + mm;
        mm = gMkV * xmk;
        const V g2 = s + mm;
        if (rowOk[jj]) {
          const size_t cs = (size_t)G.ncell * sizeof(T);
          *reinterpret_cast<V*>(po + (size_t)gcomp0 * cs + off[jj]) = g0;
          *reinterpret_cast<V*>(po + (size_t)(gcomp0 + 1) * cs + off[jj]) = g1;
          *reinterpret_cast<V*>(po + (size_t)(gcomp0 + 2) * cs + off[jj]) = g2;
        }
      } else {
        V axv;
        {
          V s = cPi * xpi;
          V mm = cCi * xc;
          s = s + mm;
          mm = cMi * xmi;
          s = s + mm;
          axv = s;
          s = cPj[jj] * dn;
          mm = cCj[jj] * xc;
          s = s + mm;
          mm = cMj[jj] * up;
          s = s + mm;
          axv = axv + s;
          s = cPkV * xpk;
          mm = cCkV * xc;
          s = s + mm;
          mm = cMkV * xmk;
          s = s + mm;
          axv = axv + s;
          if (PHASE != 3) {   // the Euler step's Laplacian: no coefficient, sign +1 (pa_tile3d_euler); x * 1 is x
            axv = axv * cf;
            axv = axv * sgn;
          }
        }
        V adv = (V)(T)0;
        if constexpr (DIV) {
          // Div(u phi) of this row, scheme KIND (fdc.py:708-772; 4 = upwind as tests/test_fdm.py:239 states it)
          const V xp3[3] = {xpi, dn, xpk}, xm3[3] = {xmi, up, xmk};
          if (KIND == 4) {
            V upl = uplC, umi = umiC;
            if constexpr (HASU) {
              const V uc = U[HC][jj];
#pragma unroll
              for (int v = 0; v < VEC; ++v) {
                upl[v] = uc[v] > (T)0 ? uc[v] : (T)0;
                umi[v] = uc[v] < (T)0 ? uc[v] : (T)0;
              }
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              V t;
              if constexpr (US == 1) {          // u >= 0: u- = +0, its half is a signed zero
                V bwd = xc - xm3[a];
                t = upl * bwd;
              } else if constexpr (US == 2) {   // u < 0: u+ = +0
                V fwd = xp3[a] - xc;
                t = umi * fwd;
              } else {
                V bwd = xc - xm3[a];
                V fwd = xp3[a] - xc;
                t = upl * bwd;
                V m2 = umi * fwd;
                t = t + m2;
              }
              t = t * A.ih[a];
              adv = adv + t;
            }
          } else if (KIND == 3) {
            V uc;
            if constexpr (HASU) uc = U[HC][jj]; else uc = (V)A.u;
            V cP, cC, cM;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
              cP[v] = (T)2 * (uc[v] < (T)0 ? uc[v] : (T)0);
              cC[v] = (T)0 * ((T)2 * uc[v]);
              cM[v] = (T)2 * (uc[v] > (T)0 ? uc[v] : (T)0);
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              V t = cP * xp3[a];
              V m2 = cC * xc;
              t = t + m2;
              m2 = cM * xm3[a];
              t = t + m2;
              adv = adv + t;
            }
          } else {
            V uc;
            if constexpr (HASU) uc = U[HC][jj]; else uc = (V)A.u;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              V cP = uc, cC = (T)0 * uc, cM = -uc;
#pragma unroll
              for (int v = 0; v < VEC; ++v) {
                const bool lo = a == 0 ? iPLo : (a == 1 ? rPLo[jj] : cPLo[v]);
                const bool hi = a == 0 ? iPHi : (a == 1 ? rPHi[jj] : cPHi[v]);
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
        }
        V res;
        if constexpr (PHASE == 3) {
          V qv = A.p0 * axv;
          qv = qv - adv;
          qv = A.p1 * qv;
          res = xc + qv;
        } else if constexpr (KIND != 0) {
          // sum_k sign_k Aop_k (ops.py:122-154) of {Laplacian, Div}: p0 = sign of the Div term, p1 != 0: Div first
          V dv = adv * A.p0;
          if (A.lap_off) axv = (V)(T)0;
          if (A.p1 != (T)0) res = dv + axv; else res = axv + dv;
        } else {
          res = axv;
        }
        if (rowOk[jj]) *reinterpret_cast<V*>(po + off[jj]) = res;
      }
    }
    };   // body
    if (rci != 0 || iPLo_ || iPHi_) body(std::false_type{}); else body(std::true_type{});
  };

  for (int q = 0; q < CI; q += 4) {
    step(I0{}, q);
    if (q + 1 < CI) step(I1{}, q + 1);
    if (q + 2 < CI) step(I2{}, q + 2);
    if (q + 3 < CI) step(I3{}, q + 3);
  }
}

// ---- host side ---------------------------------------------------------------------------------------
template <typename T, int RJ, int PHASE, int KIND, bool HASU, bool BCL = false, int US = 0>
static int sf_blocks_per_cu() {
  static int cached = 0;
  if (!cached) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_sf<T, RJ, PHASE, KIND, HASU, BCL, US>, 256, 0) != hipSuccess || n <= 0) n = 4;
    cached = n;
  }
  return cached;
}

template <typename T, int RJ, int PHASE, int KIND, bool HASU, bool BCL = false, int US = 0>
static int launch_sf(pa_ctx* c, Cg3dArgs<T>& A) {
  constexpr int VEC = VecOf<T>::N;
  constexpr int TJ = 4 * RJ, TK = 64 * VEC;
  const DevGeom& G = c->G;
  A.tiles_j = (int)((G.n1 + TJ - 1) / TJ);
  A.tiles_k = (int)((G.n2 + TK - 1) / TK);
  const int tiles = A.tiles_j * A.tiles_k;
  const int capacity = cus_of(c) * sf_blocks_per_cu<T, RJ, PHASE, KIND, HASU, BCL, US>();
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
    fprintf(stderr, "[pyapes_hip] k_sf phase %d kind %d RJ %d%s: tiles %dx%d chunks %d (CI ~%lld) blocks %d, %d blocks/CU\n",
            PHASE, KIND, RJ, BCL ? " (BC on load)" : "", A.tiles_j, A.tiles_k, chunks, (long long)(G.n0 / chunks), nblk,
            sf_blocks_per_cu<T, RJ, PHASE, KIND, HASU, BCL, US>());
  }
  hipLaunchKernelGGL((k_sf<T, RJ, PHASE, KIND, HASU, BCL, US>), dim3(nblk), dim3(256), 0, c->stream, A);
  return nblk;
}

// Rows per wave (measured, MI355X, us per launch of the upwind Div at 1 / 2 / 4 rows; k_cg3d for comparison):
//   fp32 512^3: 260 / 231 / 207 (284)   256^3: 26.4 / 25.5 / 27.3 (39.4)   128^3: 14.8 / 11.8 / 11.9 (14.7)
//   fp64 512^3: 493 / 442 / 447 (553)   256^3: 45.3 / 45.2 / 45.5 (65.3)   128^3: 14.0 / 11.4 / 12.9 (15.1)
// Four rows (least re-read of the rows above / below, ~230 VGPRs, two waves per SIMD) where a workgroup still
// marches >= 32 planes, else two (the three planes of prologue weigh less on short chunks).
template <typename T, int PHASE, int KIND>
static int launch_sf_any(pa_ctx* c, Cg3dArgs<T>& A) {
  constexpr int VEC = VecOf<T>::N;
  const DevGeom& G = c->G;
  int rj;
  if (G.n1 <= 4) {
    rj = 1;
  } else if (G.n1 <= 8) {
    rj = 2;
  } else {
    const int64_t tiles4 = ((G.n1 + 15) / 16) * ((G.n2 + 64 * VEC - 1) / (64 * VEC));
    const int64_t chunks4 = std::max<int64_t>(1, (int64_t)cus_of(c) * 2 / tiles4);
    rj = G.n0 / chunks4 >= 32 ? 4 : 2;
  }
  if constexpr (PHASE == 3 && KIND == PA_OP_DIV_UPWIND) {   // BC on load: the upwind march (BASELINE config 4)
    if (A.bcl_type[0] | A.bcl_type[1] | A.bcl_type[2] | A.bcl_type[3] | A.bcl_type[4] | A.bcl_type[5]) {
      if (rj < 2 || (c->G.n1 - 1) % rj == 0) return 0;   // PATCH: rows n1 - 2, n1 - 1 in one wave's block
      if (A.aux) return rj == 2 ? launch_sf<T, 2, 3, KIND, true, true>(c, A) : launch_sf<T, 4, 3, KIND, true, true>(c, A);
      if (PA_SF_USIGN) {   // scalar speed: its sign is a launch-time fact (US, above)
        if (A.u < (T)0) return rj == 2 ? launch_sf<T, 2, 3, KIND, false, true, 2>(c, A) : launch_sf<T, 4, 3, KIND, false, true, 2>(c, A);
        return rj == 2 ? launch_sf<T, 2, 3, KIND, false, true, 1>(c, A) : launch_sf<T, 4, 3, KIND, false, true, 1>(c, A);
      }
      return rj == 2 ? launch_sf<T, 2, 3, KIND, false, true>(c, A) : launch_sf<T, 4, 3, KIND, false, true>(c, A);
    }
  }
  if constexpr (KIND == PA_OP_DIV_UPWIND) {
    if (PA_SF_USIGN && !A.aux) {
      if (A.u < (T)0) {
        switch (rj) {
          case 1: return launch_sf<T, 1, PHASE, KIND, false, false, 2>(c, A);
          case 2: return launch_sf<T, 2, PHASE, KIND, false, false, 2>(c, A);
          default: return launch_sf<T, 4, PHASE, KIND, false, false, 2>(c, A);
        }
      }
      switch (rj) {
        case 1: return launch_sf<T, 1, PHASE, KIND, false, false, 1>(c, A);
        case 2: return launch_sf<T, 2, PHASE, KIND, false, false, 1>(c, A);
        default: return launch_sf<T, 4, PHASE, KIND, false, false, 1>(c, A);
      }
    }
  }
  constexpr bool CAN_U = (PHASE == 3 || (PHASE == 2 && KIND != 0));
  if constexpr (CAN_U) {
    if (A.aux) {
      switch (rj) {
        case 1: return launch_sf<T, 1, PHASE, KIND, true>(c, A);
        case 2: return launch_sf<T, 2, PHASE, KIND, true>(c, A);
        default: return launch_sf<T, 4, PHASE, KIND, true>(c, A);
      }
    }
  }
  switch (rj) {
    case 1: return launch_sf<T, 1, PHASE, KIND, false>(c, A);
    case 2: return launch_sf<T, 2, PHASE, KIND, false>(c, A);
    default: return launch_sf<T, 4, PHASE, KIND, false>(c, A);
  }
}

// can k_sf take this launch?  Full 16-byte vectors only (mode 1 of cg3d_mode), scalar coefficient.
template <typename T, int PHASE>
static bool sf_applies(const pa_ctx* c, const Cg3dArgs<T>& A, int mode) {
  // 32-bit byte offsets inside a plane
  return c->sf && mode == 1 && !A.coeff_f && c->G.act[0] && (size_t)c->G.s0 * sizeof(T) < ((size_t)1 << 31) &&
         (PHASE == 3 ? A.out_all != 0 : (PHASE == 2 ? A.interior_only == 0 : true));
}
