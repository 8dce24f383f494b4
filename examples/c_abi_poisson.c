/*
 * c_abi_poisson.c -- libpyapes_hip used from plain C: no Python, no PyTorch, only the HIP runtime for
 * device memory.  Shows that the drop-in boundary (include/pyapes_hip.h) is a language-neutral C ABI.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ examples/c_abi_poisson.c -Iinclude -I/opt/rocm/include \
 *       -Lpyapes_amd/lib -lpyapes_hip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/pyapes_amd/lib -Wl,-rpath,/opt/rocm/lib -lm -o /tmp/c_abi_poisson && /tmp/c_abi_poisson
 *
 * Problem 1: the reference's 3-D Poisson test (tests/test_solver.py:30-88, 3-D case): Box[0:1]^3, dx = 0.1,
 *   homogeneous Dirichlet, rhs = sin(pi x) sin(pi y) sin(pi z), CG tol 1e-6.  The rhs is an eigenvector of
 *   the discrete operator, so CG stops after 2 iterations on  x = rhs / lambda_h,
 *   lambda_h = -3 (2 - 2 cos(pi h)) / h^2.
 * Problem 2: 33^3, Dirichlet / Neumann mix of BASELINE config 5's family, rhs = sin(pi x) cos(pi y) z,
 *   CG tol 1e-10 (the reference needs 402 iterations on these inputs).
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "pyapes_hip.h"

#define CK(call)                                                                  \
  do {                                                                            \
    int rc_ = (call);                                                             \
    if (rc_ != PA_OK) {                                                           \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, pa_last_error(ctx));          \
      return 1;                                                                   \
    }                                                                             \
  } while (0)
#define HK(call)                                                                  \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_));                \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

static const double PI = 3.14159265358979323846;

static int solve(pa_ctx* ctx, int n, const int* types, const double* vals, int which_rhs, double tol,
                 pa_report* rep, double* x_host) {
  const int64_t nn[3] = {n, n, n};
  const double h = 1.0 / (n - 1);
  const double dx[3] = {h, h, h};
  const size_t cells = (size_t)n * n * n, bytes = cells * sizeof(double);
  double* rhs_h = (double*)malloc(bytes);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      for (int k = 0; k < n; ++k) {
        const double x = i * h, y = j * h, z = k * h;
        rhs_h[((size_t)i * n + j) * n + k] =
            which_rhs == 0 ? sin(PI * x) * sin(PI * y) * sin(PI * z) : sin(PI * x) * cos(PI * y) * z;
      }
  double *x_d = NULL, *rhs_d = NULL;
  HK(hipMalloc((void**)&x_d, bytes));
  HK(hipMalloc((void**)&rhs_d, bytes));
  HK(hipMemset(x_d, 0, bytes));
  HK(hipMemcpy(rhs_d, rhs_h, bytes, hipMemcpyHostToDevice));
  CK(pa_grid_set(ctx, 3, nn, dx, PA_F64, 0, n));
  for (int f = 0; f < 6; ++f)   /* faces in the factory order xl, xu, yl, yu, zl, zu; dxf = x_face - x_prev */
    CK(pa_bc_set(ctx, f, f, types[f], vals[f], NULL, (f & 1) ? h : -h));
  pa_term lap = {PA_OP_LAPLACIAN, 1, 1.0, 1.0, NULL, 0.0, NULL};   /* fdm.laplacian(1.0, var) */
  CK(pa_eq_set(ctx, 1, &lap));
  CK(pa_rhs_adjust(ctx, rhs_d));                                   /* Solver.set_eq: rhs += rhs_adj */
  CK(pa_cg(ctx, x_d, rhs_d, tol, 1000, rep));
  HK(hipMemcpy(x_host, x_d, bytes, hipMemcpyDeviceToHost));
  HK(hipFree(x_d));
  HK(hipFree(rhs_d));
  free(rhs_h);
  return 0;
}

int main(void) {
  pa_ctx* ctx = NULL;
  if (pa_ctx_create(0, NULL, &ctx) != PA_OK) {
    fprintf(stderr, "pa_ctx_create: %s\n", pa_last_error(NULL));
    return 1;
  }
  printf("%s\n", pa_version());
  pa_report rep;

  /* problem 1 */
  {
    const int n = 11;
    const int types[6] = {PA_BC_DIRICHLET, PA_BC_DIRICHLET, PA_BC_DIRICHLET, PA_BC_DIRICHLET, PA_BC_DIRICHLET, PA_BC_DIRICHLET};
    const double vals[6] = {0, 0, 0, 0, 0, 0};
    double* x = (double*)malloc(sizeof(double) * n * n * n);
    if (solve(ctx, n, types, vals, 0, 1e-6, &rep, x)) return 1;
    const double h = 1.0 / (n - 1), lam = -3.0 * (2.0 - 2.0 * cos(PI * h)) / (h * h);
    double err = 0.0;
    for (int i = 1; i < n - 1; ++i)
      for (int j = 1; j < n - 1; ++j)
        for (int k = 1; k < n - 1; ++k) {
          const double ex = sin(PI * i * h) * sin(PI * j * h) * sin(PI * k * h) / lam;
          const double d = fabs(x[((size_t)i * n + j) * n + k] - ex);
          if (d > err) err = d;
        }
    printf("poisson11: itr %lld converge %d tol %.6e max|x - rhs/lambda_h| %.3e\n", (long long)rep.itr,
           (int)rep.converge, rep.tol, err);
    free(x);
  }
  /* problem 2 */
  {
    const int n = 33;
    const int types[6] = {PA_BC_DIRICHLET, PA_BC_NEUMANN, PA_BC_DIRICHLET, PA_BC_NEUMANN, PA_BC_DIRICHLET, PA_BC_NEUMANN};
    const double vals[6] = {0.0, 0.5, 0.0, 0.0, 1.0, -0.25};
    double* x = (double*)malloc(sizeof(double) * n * n * n);
    if (solve(ctx, n, types, vals, 1, 1e-10, &rep, x)) return 1;
    printf("mixed33: itr %lld converge %d tol %.6e x[16,16,16] %.12f gpu_ms %.3f\n", (long long)rep.itr,
           (int)rep.converge, rep.tol, x[((size_t)16 * n + 16) * n + 16], rep.gpu_ms);
    free(x);
  }
  pa_ctx_destroy(ctx);
  return 0;
}
