// pa_scalar_steps.h -- the scalar steps of the solver loops on the device-resident state (shared by the
// single-block / folded kernels of pa_solver.hip and the resident small-mesh solver of pa_resident.hip)
#pragma once
#include "pa_host.h"

template <typename T>
__device__ __forceinline__ double pa_nan_to_num(T v) {
  return (isnan(v) || isinf(v)) ? 0.0 : (double)v;  // linalg.py:302-305
}

// the scalar steps of a CG iteration on the device-resident state
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

// Jacobi: stop test of one sweep (same test as CG's, linalg.py:134, 321-338)
template <typename T>
__device__ __forceinline__ void pa_logic_jacobi(SolverScalars* sc, double dx2) {
  T tol = (T)sqrt(dx2);
  sc->tol = (double)tol;
  if (isnan(tol) || isinf(tol)) { sc->err = 1; sc->done = 1; return; }
  sc->itr += 1;
  if (sc->itr > sc->max_it || !(sc->tol > sc->tolerance)) sc->done = 1;
}
