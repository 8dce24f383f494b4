// mallocbench.hip -- what an allocation costs on this part: hipMalloc / hipFree wall time by size, the rate of a
// hipMemsetAsync and of a device-to-device copy of the same block (the price list of the online placement search of
// the CG set-up, pa_solver.hip).  hipcc --offload-arch=gfx950 -O2 mallocbench.hip -o mallocbench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
int main() {
  hipStream_t s;
  hipStreamCreate(&s);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t sizes[] = {(size_t)32 << 20, (size_t)128 << 20, (size_t)512 << 20, (size_t)1 << 30, (size_t)2 << 30};
  for (size_t sz : sizes) {
    for (int rep = 0; rep < 3; ++rep) {
      void *a = nullptr, *b = nullptr;
      double t0 = now_us();
      if (hipMalloc(&a, sz) != hipSuccess) { printf("malloc failed\n"); return 1; }
      double t1 = now_us();
      hipMalloc(&b, sz);
      double t2 = now_us();
      hipEventRecord(e0, s);
      hipMemsetAsync(a, 0, sz, s);
      hipEventRecord(e1, s);
      hipEventSynchronize(e1);
      float ms_set = 0, ms_cp = 0, ms_set2 = 0;
      hipEventElapsedTime(&ms_set, e0, e1);
      hipEventRecord(e0, s);
      hipMemsetAsync(a, 0, sz, s);
      hipEventRecord(e1, s);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms_set2, e0, e1);
      hipEventRecord(e0, s);
      hipMemcpyAsync(b, a, sz, hipMemcpyDeviceToDevice, s);
      hipEventRecord(e1, s);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms_cp, e0, e1);
      double t3 = now_us();
      hipFree(a);
      double t4 = now_us();
      hipFree(b);
      double t5 = now_us();
      printf("%5zu MiB rep %d: hipMalloc %8.1f us, second %8.1f us | first memset %7.1f us, again %7.1f us (%.0f GB/s) | D2D copy %7.1f us (%.0f GB/s r+w) | hipFree %8.1f / %8.1f us\n",
             sz >> 20, rep, t1 - t0, t2 - t1, ms_set * 1e3, ms_set2 * 1e3, sz / (ms_set2 * 1e-3) / 1e9, ms_cp * 1e3,
             2.0 * sz / (ms_cp * 1e-3) / 1e9, t4 - t3, t5 - t4);
    }
  }
  // hipFree while the stream is busy: does it wait for the queued work?
  {
    void *a = nullptr, *b = nullptr, *c = nullptr;
    const size_t sz = (size_t)1 << 30;
    hipMalloc(&a, sz); hipMalloc(&b, sz); hipMalloc(&c, sz);
    for (int q = 0; q < 20; ++q) hipMemcpyAsync(b, a, sz, hipMemcpyDeviceToDevice, s);
    double t0 = now_us();
    hipFree(c);
    double t1 = now_us();
    void* d = nullptr;
    hipMalloc(&d, sz);
    double t2 = now_us();
    hipStreamSynchronize(s);
    double t3 = now_us();
    printf("busy stream (20 x 1 GiB copies queued): hipFree of an idle block %8.1f us, hipMalloc %8.1f us, then the queue drains in %8.1f us\n",
           t1 - t0, t2 - t1, t3 - t2);
    hipFree(a); hipFree(b); hipFree(d);
  }
  return 0;
}
