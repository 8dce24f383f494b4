// microbenchmark: cost of a grid-wide reduction step inside a persistent kernel (bounded spins)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#ifndef SLEEPN
#define SLEEPN 1
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bool grid_wait2(unsigned* cnt, unsigned* flag, unsigned step, unsigned G) {
  __shared__ int ok2;
  if (threadIdx.x == 0) {
    unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    int good = 0;
    if (old == (step + 1) * G - 1) {
      __hip_atomic_store(flag, step + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      good = 1;
    } else {
      for (int spin = 0; spin < (1 << 22); ++spin) {
        if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= step + 1) { good = 1; break; }
        __builtin_amdgcn_s_sleep(SLEEPN);
      }
    }
    ok2 = good;
  }
  __syncthreads();
  return ok2 != 0;
}
__device__ __forceinline__ bool grid_wait(unsigned* cnt, unsigned target) {
  // thread 0 of each block spins (bounded); everyone else waits at the barrier
  __shared__ int ok;
  if (threadIdx.x == 0) {
    int good = 0;
    for (int spin = 0; spin < (1 << 22); ++spin) {
      if (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= target) { good = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    ok = good;
  }
  __syncthreads();
  return ok != 0;
}

// MODE 0: barrier only. MODE 1: all-reduce of one double per block (write partial, arrive, wait, read all partials)
template <int MODE>
__global__ void __launch_bounds__(256) k_bar(unsigned* cnt, double* part, double* out, int iters, int* err) {
  const int G = gridDim.x, b = blockIdx.x;
  double acc = (double)(b + 1);
  __shared__ double red[4];
  for (int it = 0; it < iters; ++it) {
    double* row = part + (size_t)(it & 1) * G;
    if (MODE & 1) {
      // block partial: pretend each thread has a value; wave reduce + LDS
      double v = acc * 1e-3 + threadIdx.x;
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
      __syncthreads();
      if (threadIdx.x == 0) {
        double s = (red[0] + red[1]) + (red[2] + red[3]);
        __hip_atomic_store(row + b, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    if (MODE < 2) {
      if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      if (!grid_wait(cnt, (unsigned)(it + 1) * G)) { if (threadIdx.x == 0) *err = 1; return; }
    } else {
      if (!grid_wait2(cnt, cnt + 64, (unsigned)it, (unsigned)G)) { if (threadIdx.x == 0) *err = 1; return; }
    }
    if (MODE & 1) {
      // every block reads all partials: lanes of wave 0, fixed order
      double s = 0;
      if (threadIdx.x < 64) {
        for (int j = threadIdx.x; j < G; j += 64) s += __hip_atomic_load(row + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
        if (threadIdx.x == 0) red[0] = s;
      }
      __syncthreads();
      acc = red[0];
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) out[b] = acc;
}

int main(int argc, char** argv) {
  int iters = 2000;
  unsigned* cnt; double *part, *out; int* err;
  CK(hipMalloc(&cnt, 1024)); CK(hipMalloc(&part, 2 * 1024 * 8)); CK(hipMalloc(&out, 1024 * 8)); CK(hipMalloc(&err, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 4; ++mode)
    for (int G : {8, 16, 64, 128, 256}) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(cnt, 0, 1024)); CK(hipMemset(err, 0, 4));
        void* args[] = {&cnt, &part, &out, &iters, &err};
        CK(hipEventRecord(e0));
        void* f = mode == 0 ? (void*)k_bar<0> : mode == 1 ? (void*)k_bar<1> : mode == 2 ? (void*)k_bar<2> : (void*)k_bar<3>;
        CK(hipLaunchCooperativeKernel(f, dim3(G), dim3(256), args, 0, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int h; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
        if (rep) printf("mode %d G %3d : %.2f us per step  err %d\n", mode, G, ms * 1e3 / iters, h);
      }
    }
  return 0;
}
