// streammix.hip -- what a plain streaming kernel reaches on this part as a function of its read : write MIX, at the array
// size of BASELINE config 3 (512^3 fp64 = 1 GiB per array, far beyond the 256 MiB Infinity Cache).  The CG phases are
// such mixes with a stencil on top: phase A reads r, d and writes d' (2 : 1), phase B reads d', x, r and writes x, r in
// place (3 : 2).  Question: is phase A's lower rate (5.5-5.9 TB/s moved against phase B's 6.2-6.3)
// the kernel's or the mix's?  No halo, no LDS, no reduction here: flat grid-stride loops over 16-byte lanes.
// hipcc --offload-arch=gfx950 -O2 streammix.hip -o streammix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double V2 __attribute__((ext_vector_type(2)));

template <int R, int W, bool INPLACE>
__global__ void __launch_bounds__(256) k_mix(V2* __restrict__ a0, V2* __restrict__ a1, V2* __restrict__ a2, V2* __restrict__ a3,
                                             V2* __restrict__ a4, size_t n, double beta, double* sink) {
  V2 acc = {0.0, 0.0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    V2 v = a0[i];
    if (R >= 2) v = v + beta * a1[i];
    if (R >= 3) v = v + beta * a2[i];
    if (W == 0) acc += v;
    if (INPLACE) {               // writes go back into the arrays that were read (phase B: x, r)
      if (W >= 1) a1[i] = v;
      if (W >= 2) a2[i] = v * beta;
    } else {
      if (W >= 1) a3[i] = v;
      if (W >= 2) a4[i] = v * beta;
    }
  }
  if (W == 0 && acc.x + acc.y == 12345.678) sink[0] = acc.x;   // keeps the loads alive
}

// the same mixes with the CG kernels' access pattern: a workgroup owns a tile of 16 rows x 128 doubles (4 waves x 4 rows,
// a wave covers 1 KiB of a row) and marches over a chunk of planes; the grid is one resident wave of workgroups
template <int R, int W, bool INPLACE, bool NT>
__global__ void __launch_bounds__(256) k_march(double* __restrict__ a0, double* __restrict__ a1, double* __restrict__ a2,
                                               double* __restrict__ a3, double* __restrict__ a4, int n, int chunks, double beta,
                                               int rev = 0) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tiles_k = n / 128, tiles_j = n / 16, tiles = tiles_j * tiles_k;
  // blocks b and b + 8 share an XCD: consecutive tiles of one chunk on one XCD (the CG kernels' map)
  const int nb = gridDim.x, per = nb / 8, vb = (blockIdx.x % 8) * per + blockIdx.x / 8;
  const int chunk = vb / tiles, tl = vb % tiles, tj = tl / tiles_k, tk = tl % tiles_k;
  const int i0 = (long)chunk * n / chunks, i1 = (long)(chunk + 1) * n / chunks;
  for (int q = i0; q < i1; ++q) {
    const int i = rev ? i1 - 1 - (q - i0) : q;
    V2 v[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const size_t o = ((size_t)i * n + tj * 16 + wv * 4 + jj) * n + tk * 128 + lane * 2;
      auto ld = [](const double* p) { return NT ? __builtin_nontemporal_load((const V2*)p) : *(const V2*)p; };
      v[jj] = ld(a0 + o);
      if (R >= 2) v[jj] = v[jj] + beta * ld(a1 + o);
      if (R >= 3) v[jj] = v[jj] + beta * ld(a2 + o);
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const size_t o = ((size_t)i * n + tj * 16 + wv * 4 + jj) * n + tk * 128 + lane * 2;
      auto st = [](double* p, V2 val) { if (NT) __builtin_nontemporal_store(val, (V2*)p); else *(V2*)p = val; };
      if (W >= 1) st((INPLACE ? a1 : a3) + o, v[jj]);
      if (W >= 2) st((INPLACE ? a2 : a4) + o, v[jj] * beta);
    }
  }
}

int main(int argc, char** argv) {
  const size_t N = (size_t)512 * 512 * 512, n = N / 2;
  // argv[1]: byte offset between the starts of consecutive arrays modulo their size (0: every array starts on the same
  // channel / bank phase, what five hipMalloc blocks of 1 GiB give; the library staggers its blocks by 69888 B)
  const size_t stagger = argc > 1 ? (size_t)atol(argv[1]) : 0;
  printf("stagger %zu B\n", stagger);
  V2* a[5];
  for (int q = 0; q < 5; ++q) {
    char* p;
    hipMalloc((void**)&p, N * 8 + 5 * stagger);
    hipMemset(p, 0, N * 8 + 5 * stagger);
    a[q] = (V2*)(p + q * stagger);
  }
  double* sink; hipMalloc(&sink, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, int passes, auto f) {
    for (int w = 0; w < 3; ++w) f();
    hipEventRecord(e0); for (int r = 0; r < 20; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-46s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, (double)passes * N * 8 / ms / 1e9);
  };
  char nm[128];
#define MIX(R, W, IP, label)                                                                                         \
  for (int g : {2048, 4096, 8192}) {                                                                                  \
    sprintf(nm, "%s (grid %d)", label, g);                                                                            \
    run(nm, R + W, [&] { hipLaunchKernelGGL((k_mix<R, W, IP>), dim3(g), dim3(256), 0, 0, a[0], a[1], a[2], a[3], a[4], n, 0.5, sink); }); \
  }
  MIX(1, 0, false, "1 read")
  MIX(2, 0, false, "2 reads")
  MIX(1, 1, false, "1 read : 1 write (copy)")
  MIX(2, 1, false, "2 reads : 1 write (phase A's mix)")
  MIX(3, 1, false, "3 reads : 1 write")
  MIX(2, 2, false, "2 reads : 2 writes")
  MIX(3, 2, false, "3 reads : 2 writes, five arrays")
  MIX(3, 2, true, "3 reads : 2 writes in place (phase B's mix)")
#define MARCH(R, W, IP, NT, label)                                                                                        \
  for (int cap : {1024, 2048}) {                                                                                       \
    const int tiles = (512 / 16) * (512 / 128), chunks = cap / tiles;                                                  \
    sprintf(nm, "march: %s (%d blocks)", label, tiles * chunks);                                                       \
    run(nm, R + W, [&] { hipLaunchKernelGGL((k_march<R, W, IP, NT>), dim3(tiles * chunks), dim3(256), 0, 0, (double*)a[0], (double*)a[1], \
                                            (double*)a[2], (double*)a[3], (double*)a[4], 512, chunks, 0.5); });        \
  }
  MARCH(1, 1, false, false, "1 : 1 (copy)")
  MARCH(2, 1, false, false, "2 : 1 (phase A's mix)")
  MARCH(3, 2, false, false, "3 : 2, five arrays")
  MARCH(3, 2, true, false, "3 : 2 in place (phase B's mix)")
  // with the non-temporal loads / stores the CG kernels use on their once-touched streams
  MARCH(1, 1, false, true, "nt 1 : 1 (copy)")
  MARCH(2, 1, false, true, "nt 2 : 1 (phase A's mix)")
  MARCH(3, 2, false, true, "nt 3 : 2, five arrays")
  MARCH(3, 2, true, true, "nt 3 : 2 in place (phase B's mix)")
  // the CG iteration's traffic without its stencil: A' reads r (a0), d (a1), writes d' (a2); B' reads d' (a2), x (a3), r (a0)
  // and writes x, r in place -- B' marching its chunks forward like A', or backward (what the library does: the planes
  // A' touched last, still in the 256 MiB Infinity Cache, are the ones B' reads first, and vice versa)
  for (int rev = 0; rev < 2; ++rev)
    for (int cap : {1024, 2048}) {
      const int tiles = (512 / 16) * (512 / 128), chunks = cap / tiles;
      sprintf(nm, "march nt: pair A' (2 : 1) + B' (3 : 2 in place) %s (%d blocks)", rev ? "B' BACKWARD" : "both forward", tiles * chunks);
      run(nm, 8, [&] {
        hipLaunchKernelGGL((k_march<2, 1, false, true>), dim3(tiles * chunks), dim3(256), 0, 0, (double*)a[0], (double*)a[1],
                           (double*)nullptr, (double*)a[2], (double*)nullptr, 512, chunks, 0.5, 0);
        hipLaunchKernelGGL((k_march<3, 2, true, true>), dim3(tiles * chunks), dim3(256), 0, 0, (double*)a[2], (double*)a[3],
                           (double*)a[0], (double*)nullptr, (double*)nullptr, 512, chunks, 0.5, rev);
      });
    }
  return 0;
}
