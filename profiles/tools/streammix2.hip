// streammix2.hip -- follow-up to streammix.hip.  That tool showed (profiles/r04_streammix.txt) that the CG phases run at
// the rate of their bare read : write mixes in marching form (2 : 1 = 552 us against phase A's 563; 3 : 2 in place = 860
// against phase B's 850), and that WRITES IN PLACE are cheaper than writes into a separate array (3 : 2 in place 6.24 TB/s,
// 3 : 2 into five arrays 5.54).  Question here: does phase A's output d' get cheaper when it lands NEXT TO the array it is
// computed from -- in place (not available to the real kernel: neighbours still read the old halo), or in a buffer that
// is interleaved with d row by row / plane by plane (available: a pitch) -- and what does phase B pay for reading d' from
// such an interleaved buffer?
// hipcc --offload-arch=gfx950 -O2 streammix2.hip -o streammix2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double V2 __attribute__((ext_vector_type(2)));

// element offset of (i, j, k) in slot s of a buffer interleaved with granularity MODE: 0 = two separate arrays (slot
// stride = array), 1 = rows interleaved ([i][j][slot][k]), 2 = planes interleaved ([i][slot][j][k]),
// 3 = 16-row tiles interleaved ([i][j / 16][slot][j % 16][k])
template <int MODE>
__device__ __forceinline__ size_t il_off(int i, int j, int k, int s, int n) {
  if (MODE == 1) return (((size_t)i * n + j) * 2 + s) * n + k;
  if (MODE == 2) return (((size_t)i * 2 + s) * n + j) * n + k;
  if (MODE == 3) return ((((size_t)i * (n / 16) + j / 16) * 2 + s) * 16 + (j % 16)) * n + k;
  return (size_t)s * n * n * n + ((size_t)i * n + j) * n + k;
}

// phase A's traffic: reads r (own array) and d (slot 0 of the pair), writes d' into slot 1 of the pair -- or in place
// over d (INPLACE).  NT: the non-temporal store the library uses for d'.
template <int MODE, bool INPLACE>
__global__ void __launch_bounds__(256) k_a(const double* __restrict__ r, double* dd, int n, int chunks, double beta) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tiles_k = n / 128, tiles_j = n / 16, tiles = tiles_j * tiles_k;
  const int nb = gridDim.x, per = nb / 8, vb = (blockIdx.x % 8) * per + blockIdx.x / 8;
  const int chunk = vb / tiles, tl = vb % tiles, tj = tl / tiles_k, tk = tl % tiles_k;
  const int i0 = (long)chunk * n / chunks, i1 = (long)(chunk + 1) * n / chunks;
  for (int i = i0; i < i1; ++i) {
    V2 v[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = tj * 16 + wv * 4 + jj, k = tk * 128 + lane * 2;
      const size_t o = ((size_t)i * n + j) * n + k;
      v[jj] = *(const V2*)(r + o) + beta * *(const V2*)(dd + il_off<MODE>(i, j, k, 0, n));
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = tj * 16 + wv * 4 + jj, k = tk * 128 + lane * 2;
      __builtin_nontemporal_store(v[jj], (V2*)(dd + il_off<MODE>(i, j, k, INPLACE ? 0 : 1, n)));
    }
  }
}

// phase B's traffic: reads d' (slot 1 of the pair), x, r; writes x, r in place; marches backwards
template <int MODE>
__global__ void __launch_bounds__(256) k_b(const double* dd, double* x, double* r, int n, int chunks, double alpha) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tiles_k = n / 128, tiles_j = n / 16, tiles = tiles_j * tiles_k;
  const int nb = gridDim.x, per = nb / 8, vb = (blockIdx.x % 8) * per + blockIdx.x / 8;
  const int chunk = vb / tiles, tl = vb % tiles, tj = tl / tiles_k, tk = tl % tiles_k;
  const int i0 = (long)chunk * n / chunks, i1 = (long)(chunk + 1) * n / chunks;
  for (int i = i1 - 1; i >= i0; --i) {
    V2 a[4], b[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = tj * 16 + wv * 4 + jj, k = tk * 128 + lane * 2;
      const size_t o = ((size_t)i * n + j) * n + k;
      const V2 d = *(const V2*)(dd + il_off<MODE>(i, j, k, 1, n));
      a[jj] = __builtin_nontemporal_load((const V2*)(x + o)) + alpha * d;
      b[jj] = __builtin_nontemporal_load((const V2*)(r + o)) - alpha * d;
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = tj * 16 + wv * 4 + jj, k = tk * 128 + lane * 2;
      const size_t o = ((size_t)i * n + j) * n + k;
      __builtin_nontemporal_store(a[jj], (V2*)(x + o));
      __builtin_nontemporal_store(b[jj], (V2*)(r + o));
    }
  }
}


// BiCGSTAB's x / r update (k_bicg_x, SRV form): reads x, p, r, t, v; writes x, r in place and p' into a sixth array.  Three
// traversals of the same 8 passes: FLAT grid-stride (the library's kernel), CHUNK = every block a contiguous range of
// vectors (XCD-aware, optionally backwards), MARCH = the CG kernels' tiles marching over chunks of planes.
template <bool NT>
__device__ __forceinline__ void xr_cell(size_t o, double* x, const double* p, double* r, const double* t, const double* v, double* pn,
                                        double al, double om, double be) {
  auto ld = [](const double* q) { return NT ? __builtin_nontemporal_load((const V2*)q) : *(const V2*)q; };
  const V2 pv = *(const V2*)(p + o), vv = *(const V2*)(v + o);
  const V2 sv = ld(r + o) - al * vv;
  const V2 tv = ld(t + o);
  const V2 xn = ld(x + o) + al * pv + om * sv;
  const V2 rn = sv - om * tv;
  const V2 pq = rn + be * (pv - om * vv);
  if (NT) { __builtin_nontemporal_store(xn, (V2*)(x + o)); __builtin_nontemporal_store(rn, (V2*)(r + o)); }
  else { *(V2*)(x + o) = xn; *(V2*)(r + o) = rn; }
  *(V2*)(pn + o) = pq;
}
template <bool NT>
__global__ void __launch_bounds__(256) k_x_flat(double* x, const double* p, double* r, const double* t, const double* v, double* pn, size_t nvec) {
  for (size_t iv = (size_t)blockIdx.x * 256 + threadIdx.x; iv < nvec; iv += (size_t)gridDim.x * 256)
    xr_cell<NT>(iv * 2, x, p, r, t, v, pn, 0.5, 0.25, 0.125);
}
template <bool NT, bool REV, bool XCD>
__global__ void __launch_bounds__(256) k_x_chunk(double* x, const double* p, double* r, const double* t, const double* v, double* pn, size_t nvec) {
  const int nb = gridDim.x, per = nb / 8, vb = XCD ? (blockIdx.x % 8) * per + blockIdx.x / 8 : blockIdx.x;
  const size_t L = nvec / nb, b0 = (size_t)vb * L;
  for (size_t q = threadIdx.x; q < L; q += 256) {
    const size_t iv = REV ? b0 + (L - 256 - (q - threadIdx.x)) + threadIdx.x : b0 + q;
    xr_cell<NT>(iv * 2, x, p, r, t, v, pn, 0.5, 0.25, 0.125);
  }
}
template <bool NT, bool REV>
__global__ void __launch_bounds__(256) k_x_march(double* x, const double* p, double* r, const double* t, const double* v, double* pn, int n, int chunks) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tiles_k = n / 128, tiles_j = n / 16, tiles = tiles_j * tiles_k;
  const int nb = gridDim.x, per = nb / 8, vb = (blockIdx.x % 8) * per + blockIdx.x / 8;
  const int chunk = vb / tiles, tl = vb % tiles, tj = tl / tiles_k, tk = tl % tiles_k;
  const int i0 = (long)chunk * n / chunks, i1 = (long)(chunk + 1) * n / chunks;
  for (int q = i0; q < i1; ++q) {
    const int i = REV ? i1 - 1 - (q - i0) : q;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      xr_cell<NT>(((size_t)i * n + tj * 16 + wv * 4 + jj) * n + tk * 128 + lane * 2, x, p, r, t, v, pn, 0.5, 0.25, 0.125);
  }
}

int main() {
  const int n = 512;
  const size_t N = (size_t)n * n * n;
  double *r, *x, *dd;
  hipMalloc((void**)&r, N * 8); hipMalloc((void**)&x, N * 8); hipMalloc((void**)&dd, 2 * N * 8);
  hipMemset(r, 0, N * 8); hipMemset(x, 0, N * 8); hipMemset(dd, 0, 2 * N * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, int passes, auto f) {
    for (int w = 0; w < 3; ++w) f();
    (void)hipEventRecord(e0); for (int q = 0; q < 20; ++q) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-72s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, (double)passes * N * 8 / ms / 1e9);
  };
  const int tiles = (n / 16) * (n / 128), chunks = 1024 / tiles, g = tiles * chunks;
#define A(MODE, IP, label) run("A' 2 : 1 " label, 3, [&] { hipLaunchKernelGGL((k_a<MODE, IP>), dim3(g), dim3(256), 0, 0, r, dd, n, chunks, 0.5); });
#define B(MODE, label) run("B' 3 : 2 in place, d' read from " label, 5, [&] { hipLaunchKernelGGL((k_b<MODE>), dim3(g), dim3(256), 0, 0, dd, x, r, n, chunks, 0.5); });
#define AB(MODE, label) run("pair A' + B', d / d' " label, 8, [&] { \
    hipLaunchKernelGGL((k_a<MODE, false>), dim3(g), dim3(256), 0, 0, r, dd, n, chunks, 0.5); \
    hipLaunchKernelGGL((k_b<MODE>), dim3(g), dim3(256), 0, 0, dd, x, r, n, chunks, 0.5); });
  for (int rep = 0; rep < 2; ++rep) {
    A(0, false, "d' into a separate array (the library)")
    A(0, true, "d' IN PLACE over d (hazard in the real kernel)")
    A(1, false, "d / d' interleaved row by row")
    A(3, false, "d / d' interleaved by 16-row tiles")
    A(2, false, "d / d' interleaved plane by plane")
    B(0, "a separate array (the library)")
    B(1, "the row-interleaved pair")
    B(3, "the tile-interleaved pair")
    B(2, "the plane-interleaved pair")
    AB(0, "separate arrays (the library)")
    AB(1, "interleaved row by row")
    AB(3, "interleaved by 16-row tiles")
    AB(2, "interleaved plane by plane")
  }
  {
    double *p, *t, *v, *pn;
    hipMalloc((void**)&p, N * 8); hipMalloc((void**)&t, N * 8); hipMalloc((void**)&v, N * 8); hipMalloc((void**)&pn, N * 8);
    hipMemset(p, 0, N * 8); hipMemset(t, 0, N * 8); hipMemset(v, 0, N * 8); hipMemset(pn, 0, N * 8);
    const size_t nvec = N / 2;
    for (int rep = 0; rep < 2; ++rep) {
      for (int gsz : {2048, 4096, 8192}) {
        char nm[128]; sprintf(nm, "x / r update 5 : 3, FLAT grid-stride, %d blocks (the library)", gsz);
        run(nm, 8, [&] { hipLaunchKernelGGL((k_x_flat<false>), dim3(gsz), dim3(256), 0, 0, x, p, r, t, v, pn, nvec); });
      }
      run("x / r update 5 : 3, FLAT grid-stride nt, 2048 blocks", 8, [&] { hipLaunchKernelGGL((k_x_flat<true>), dim3(2048), dim3(256), 0, 0, x, p, r, t, v, pn, nvec); });
      for (int gsz : {1024, 2048}) {
        char nm[128];
        sprintf(nm, "x / r update 5 : 3, CHUNK forward, %d blocks", gsz);
        run(nm, 8, [&] { hipLaunchKernelGGL((k_x_chunk<false, false, false>), dim3(gsz), dim3(256), 0, 0, x, p, r, t, v, pn, nvec); });
        sprintf(nm, "x / r update 5 : 3, CHUNK forward xcd, %d blocks", gsz);
        run(nm, 8, [&] { hipLaunchKernelGGL((k_x_chunk<false, false, true>), dim3(gsz), dim3(256), 0, 0, x, p, r, t, v, pn, nvec); });
        sprintf(nm, "x / r update 5 : 3, CHUNK backward xcd nt, %d blocks", gsz);
        run(nm, 8, [&] { hipLaunchKernelGGL((k_x_chunk<true, true, true>), dim3(gsz), dim3(256), 0, 0, x, p, r, t, v, pn, nvec); });
        const int ch = gsz / tiles;
        sprintf(nm, "x / r update 5 : 3, MARCH forward, %d blocks", tiles * ch);
        run(nm, 8, [&] { hipLaunchKernelGGL((k_x_march<false, false>), dim3(tiles * ch), dim3(256), 0, 0, x, p, r, t, v, pn, n, ch); });
        sprintf(nm, "x / r update 5 : 3, MARCH backward nt, %d blocks", tiles * ch);
        run(nm, 8, [&] { hipLaunchKernelGGL((k_x_march<true, true>), dim3(tiles * ch), dim3(256), 0, 0, x, p, r, t, v, pn, n, ch); });
      }
    }
  }
  return 0;
}
