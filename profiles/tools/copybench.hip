// practical ceilings for a 1-read + 1-write streaming kernel on this part, by access pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float V4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_flat(const V4* __restrict__ a, V4* __restrict__ b, size_t n, int nt) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    V4 v = nt & 1 ? __builtin_nontemporal_load(a + i) : a[i];
    if (nt & 2) __builtin_nontemporal_store(v, b + i); else b[i] = v;
  }
}
// each block owns a contiguous range (like a marching chunk), 4 loads in flight per thread
__global__ void __launch_bounds__(256) k_range(const V4* __restrict__ a, V4* __restrict__ b, size_t n, int nt) {
  size_t per = (n + gridDim.x - 1) / gridDim.x;
  size_t lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += 1024) {
    V4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + u * 256 < hi) v[u] = nt & 1 ? __builtin_nontemporal_load(a + i + u * 256) : a[i + u * 256];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + u * 256 < hi) { if (nt & 2) __builtin_nontemporal_store(v[u], b + i + u * 256); else b[i + u * 256] = v[u]; }
  }
}
// tile-marching: block = tile of TJ rows x 256 floats, marches over CI planes of an n^3 array
__global__ void __launch_bounds__(256) k_march(const float* __restrict__ a, float* __restrict__ b, int n, int chunks, int rj, int nt) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tiles_k = n / 256, tiles_j = n / (4 * rj), tiles = tiles_j * tiles_k;
  const int chunk = blockIdx.x / tiles, tl = blockIdx.x % tiles, tj = tl / tiles_k, tk = tl % tiles_k;
  const int i0 = (long)chunk * n / chunks, i1 = (long)(chunk + 1) * n / chunks;
  for (int i = i0; i < i1; ++i)
    for (int jj = 0; jj < rj; ++jj) {
      size_t o = ((size_t)i * n + tj * 4 * rj + wv * rj + jj) * n + tk * 256 + lane * 4;
      V4 v = nt & 1 ? __builtin_nontemporal_load((const V4*)(a + o)) : *(const V4*)(a + o);
      if (nt & 2) __builtin_nontemporal_store(v, (V4*)(b + o)); else *(V4*)(b + o) = v;
    }
}
int main() {
  const int n = 512; const size_t N = (size_t)n * n * n; float *a, *b;
  hipMalloc(&a, N * 4); hipMalloc(&b, N * 4); hipMemset(a, 1, N * 4); hipMemset(b, 0, N * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto f) {
    for (int w = 0; w < 3; ++w) f();
    hipEventRecord(e0); for (int r = 0; r < 20; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-44s %8.1f us  %6.2f TB/s (read+write)\n", name, ms * 1e3, 2.0 * N * 4 / ms / 1e9);
  };
  char nm[128];
  for (int nt = 0; nt < 4; ++nt) for (int g : {1024, 2048, 4096, 8192}) {
    sprintf(nm, "flat grid %d nt=%d", g, nt); run(nm, [&] { hipLaunchKernelGGL(k_flat, dim3(g), dim3(256), 0, 0, (const V4*)a, (V4*)b, N / 4, nt); });
  }
  for (int nt = 0; nt < 4; ++nt) for (int g : {512, 1024, 2048}) {
    sprintf(nm, "range grid %d nt=%d", g, nt); run(nm, [&] { hipLaunchKernelGGL(k_range, dim3(g), dim3(256), 0, 0, (const V4*)a, (V4*)b, N / 4, nt); });
  }
  for (int nt = 0; nt < 4; ++nt) for (int rj : {1, 2, 4}) for (int cap : {512, 1024, 2048}) {
    int tiles = (n / 256) * (n / (4 * rj)); int chunks = cap / tiles; if (chunks < 1) chunks = 1;
    sprintf(nm, "march rj %d blocks %d (chunks %d) nt=%d", rj, tiles * chunks, chunks, nt);
    run(nm, [&] { hipLaunchKernelGGL(k_march, dim3(tiles * chunks), dim3(256), 0, 0, a, b, n, chunks, rj, nt); });
  }
  run("hipMemcpyDtoD", [&] { hipMemcpyAsync(b, a, N * 4, hipMemcpyDeviceToDevice, 0); });
  return 0;
}
