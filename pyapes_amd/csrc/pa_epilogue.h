// pa_epilogue.h -- "last block finishes the reduction": the block of a kernel that takes the last
// ticket sums the per-block partial rows (same fixed order as k_cg_post_a / k_cg_post_b, so the
// result is bit-identical to the separate reduction kernels) and, on a single GPU, runs the scalar
// logic of the iteration, which removes the two single-block launches per CG iteration.
// OPT-IN (PYAPES_HIP_EPILOGUE=1), because on MI355X it loses: each of the 8 XCDs has its own L2, so
// the agent-scope release / acquire fences this pattern needs write back and invalidate an L2 --
// measured +25 us per epilogue whatever the grid size (256^3 fp64: 0.254 vs 0.195 ms / iteration;
// 64x64x128: 0.071 vs 0.027), against 4.7 us for the single-block kernel it replaces, whose
// visibility comes for free with the kernel boundary.  Kept as the measured alternative and as a
// bit-exact cross-check of the reduction path (tests/test_gpu_epilogue.py).
#pragma once
#include <hip/hip_runtime.h>

#include "pa_host.h"


// same summation tree as pa_reduce_partials (pa_core.hip).  Plain (pipelined) loads: the caller has
// passed an agent-scope acquire fence after taking the last ticket, which invalidates this CU's L1, and
// every other block released its row before its ticket -- per-element atomic loads would serialise
// (measured: +50 us on 2048 rows).
__device__ __forceinline__ double pa_reduce_partials_coh(const double* partials, int nblk, int ns, int s,
                                                         double* sm) {
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
__device__ __forceinline__ void pa_logic_a(SolverScalars* sc, const double* sums) {  // linalg.py:118-120
  T dAd = (T)sums[0];
  T rr = (T)sc->rr;
  sc->dAd = (double)dAd;
  T a = rr / dAd;
  sc->alpha = (isnan(a) || isinf(a)) ? 0.0 : (double)a;
}

template <typename T>
__device__ __forceinline__ void pa_logic_b(SolverScalars* sc, const double* sums) {  // linalg.py:128-141, 321-338
  T rr_new = (T)sums[1];
  T tol = (T)sqrt(sums[2]);
  sc->tol = (double)tol;
  if (isnan(tol) || isinf(tol)) {  // linalg.py:334-336 raises before beta / itr
    sc->err = 1;
    sc->done = 1;
    return;
  }
  T rr_old = (T)sc->rr;
  sc->rr_old = (double)rr_old;
  sc->beta = (double)(rr_new / rr_old);
  sc->rr = (double)rr_new;
  sc->itr += 1;
  if (sc->itr > sc->max_it || !(sc->tol > sc->tolerance)) sc->done = 1;
}

// Call once per block after the block's own partial row has been stored by thread 0.
template <typename T>
__device__ __forceinline__ void pa_cg_epilogue(const CgEpi& E) {
  if (E.kind == 0) return;
  __shared__ int is_last;
  __shared__ double sm[16];
  if (threadIdx.x == 0) {
    __threadfence();                                  // my partial row is visible device-wide ...
    unsigned int t = atomicAdd(E.ticket, 1u);         // ... before my ticket is
    is_last = (t == gridDim.x - 1);
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  const int npart = E.npart < 0 ? (int)gridDim.x : E.npart;
  if (E.kind == 1) {
    double v = pa_reduce_partials_coh(E.part, npart, 1, 0, sm);
    if (threadIdx.x == 0) {
      E.sums[0] = v;
      if (E.logic) pa_logic_a<T>(E.sc, E.sums);
    }
  } else {
    double rr = pa_reduce_partials_coh(E.part, npart, 2, 0, sm);
    double dx2 = pa_reduce_partials_coh(E.part, npart, 2, 1, sm);
    double sh = E.nshell > 0 ? pa_reduce_partials_coh(E.part_shell, E.nshell, 1, 0, sm) : 0.0;
    if (threadIdx.x == 0) {
      E.sums[1] = rr;
      E.sums[2] = dx2 + sh;
      if (E.logic) pa_logic_b<T>(E.sc, E.sums);
    }
  }
  if (threadIdx.x == 0) *E.ticket = 0u;               // ready for the next launch
}
