// pa_rfp.hip -- the explicit operators either side of the stencil path that the reference's
// Fokker-Planck users need: the general Div (Jac advection, vector target,
// edge=True in n-D), DiffFlux, and the rz-only Friction / Diffusion of solver/rfp.py.
// Every expression keeps the reference's operation order (one rounding per product / sum /
// quotient; the library is built with -ffp-contract=off), so the outputs are bit-exact.
#include <hip/hip_runtime.h>

#include <cstring>

#include "../../include/pyapes_hip.h"
#include "pa_device.h"
#include "pa_host.h"

namespace {

template <typename T>
__device__ __forceinline__ T nn(T v) {  // torch.nan_to_num(v, nan=0, posinf=0, neginf=0)
  return (isnan(v) || isinf(v)) ? (T)0 : v;
}

template <typename T>
struct DivSpec {
  const T* x[3];   // target component per internal axis
  const T* ui[3];  // advection of the interior stencil per internal axis (null: scalar u)
  const T* ue[3];  // advection of the edge formula per internal axis (null: scalar u)
  T u;
  int kind, edge;
};

// fdc.py:93-102 (sum over axes of coefficient[a][a] applied to component a), 708-772 (coefficients),
// 290-361 (edge planes of each axis' contribution)
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_div_general(DevGeom G, DevEq<T> E, DivSpec<T> S, T* __restrict__ y) {
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < G.ncell; o += (int64_t)gridDim.x * blockDim.x) {
    int64_t i, j, k;
    pa_decode(G, o, i, j, k);
    const int64_t g[3] = {i, j, k};
    const int64_t N[3] = {G.n0, G.n1, G.n2};
    const int64_t st[3] = {G.s0, G.s1, 1};
    T out = (T)0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (!G.act[a]) continue;
      const T* x = S.x[a];
      const int64_t ga = g[a], Na = N[a];
      const T xc = x[o];
      T disc;
      if (S.edge && (ga == 0 || ga == Na - 1)) {
        const bool lower = ga == 0;
        const int64_t d1 = lower ? st[a] : -st[a];
        const T t0 = xc, t1 = x[o + d1], t2 = x[o + 2 * d1];
        T e = (T)1.5 * t0;
        T m = (T)2 * t1;
        e = e - m;
        m = (T)0.5 * t2;
        e = e + m;
        if (lower) e = -e;
        e = e / E.grd.h[a];
        const T adv = S.ue[a] ? S.ue[a][o] : S.u;
        disc = e * adv;
        if (E.rz && a == PA_RZ_AXIS) {  // fdc.py:330-334 (no advection factor) / 350-357
          const T r = E.rz[5 * E.rz_n + ga];
          T add = lower ? t0 : t0 * adv;
          add = add / r;
          disc = disc + nn(add);
        }
      } else {
        const int64_t op = o + (pa_wrap(ga + 1, Na) - ga) * st[a];
        const int64_t om = o + (pa_wrap(ga - 1, Na) - ga) * st[a];
        const T xp = x[op], xm = x[om];
        const T* uf = S.ui[a];
        const T ucen = uf ? uf[o] : S.u;
        if (S.kind == PA_OP_DIV_CENTRAL) {
          T cP = uf ? uf[op] : S.u;
          T cM = -(uf ? uf[om] : S.u);
          T cC = (T)0 * ucen;
          if (E.rz && a == PA_RZ_AXIS) cC = E.rz[4 * E.rz_n + ga] * ucen;
          if (G.bct[2 * a] == 4 && ga == 1) cM = (T)0;
          if (G.bct[2 * a + 1] == 4 && ga == Na - 2) cP = (T)0;
          cP = cP / E.grd.h2[a];
          cC = cC / E.grd.h2[a];
          cM = cM / E.grd.h2[a];
          T s = cP * xp;
          T m = cC * xc;
          s = s + m;
          m = cM * xm;
          disc = s + m;
        } else if (S.kind == PA_OP_DIV_UPWIND_COMPAT) {
          T cP = (T)2 * (ucen < (T)0 ? ucen : (T)0);
          T cC = (T)0 * ((T)2 * ucen);
          if (E.rz && a == PA_RZ_AXIS) cC = E.rz[4 * E.rz_n + ga] * ((T)2 * ucen);
          T cM = (T)2 * (ucen > (T)0 ? ucen : (T)0);
          T s = cP * xp;
          T m = cC * xc;
          s = s + m;
          m = cM * xm;
          disc = s + m;
        } else {
          T upl = ucen > (T)0 ? ucen : (T)0;
          T umi = ucen < (T)0 ? ucen : (T)0;
          T bwd = xc - xm;
          T fwd = xp - xc;
          T s = upl * bwd;
          T m = umi * fwd;
          s = s + m;
          disc = s * E.grd.ih[a];
          if (E.rz && a == PA_RZ_AXIS) {
            T cC = E.rz[4 * E.rz_n + ga] * ucen;
            cC = cC / E.grd.h2[a];
            m = cC * xc;
            disc = disc + m;
          }
        }
      }
      out = out + disc;
    }
    y[o] = out;
  }
}

// fdc.py:818-856: out[i] = sum_j w_i D_ij J_j, w = r for the r row of an rz mesh
template <typename T>
struct FluxArgs {
  const T* D[9];
  const T* J[3];
  int nd;
};

template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_diff_flux(DevGeom G, DevEq<T> E, FluxArgs<T> A, T* __restrict__ out) {
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < G.ncell; o += (int64_t)gridDim.x * blockDim.x) {
    int64_t i, j, k;
    pa_decode(G, o, i, j, k);
    for (int p = 0; p < A.nd; ++p) {
      T acc = (T)0;
      for (int q = 0; q < A.nd; ++q) {
        T d = A.D[p * A.nd + q][o];
        if (E.rz && p == 0) d = E.rz[5 * E.rz_n + j] * d;   // 2-D: mesh axis 0 = internal axis 1 = index j
        T m = d * A.J[q][o];
        acc = acc + m;
      }
      out[(int64_t)p * G.ncell + o] = acc;
    }
  }
}

// ---- solver/rfp.py: Friction (19-82) and Diffusion (85-218), rz meshes ---------------------------
template <typename T>
struct Rz2 {
  int64_t nr, nz;
  const T* R;  // r coordinate per r node
  __device__ __forceinline__ T at(const T* t, int64_t i, int64_t k, int di, int dk) const {
    return t[pa_wrap(i + di, nr) * nz + pa_wrap(k + dk, nz)];
  }
};

template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_rfp_friction(DevGeom G, DevEq<T> E, const T* __restrict__ Hr,
                                                            const T* __restrict__ Hz, const T* __restrict__ f,
                                                            T* __restrict__ out) {
  const Rz2<T> M{G.n1, G.n2, E.rz + 5 * E.rz_n};
  const T dr = E.grd.h[1], dz = E.grd.h[2];
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < G.ncell; o += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = o / M.nz, k = o - i * M.nz;
    const T hr = Hr[o], hz = Hz[o], p = f[o];
    const T Arp = (M.at(Hr, i, k, 1, 0) + hr) / (T)2, Arm = (hr + M.at(Hr, i, k, -1, 0)) / (T)2;
    const T Azp = (M.at(Hz, i, k, 0, 1) + hz) / (T)2, Azm = (hz + M.at(Hz, i, k, 0, -1)) / (T)2;
    const T Prp = (M.at(f, i, k, 1, 0) + p) / (T)2, Prm = (p + M.at(f, i, k, -1, 0)) / (T)2;
    const T Pzp = (M.at(f, i, k, 0, 1) + p) / (T)2, Pzm = (p + M.at(f, i, k, 0, -1)) / (T)2;
    const T r = M.R[i];
    const T r_p = (M.R[pa_wrap(i + 1, M.nr)] + r) / (T)2, r_m = (r + M.R[pa_wrap(i - 1, M.nr)]) / (T)2;
    const T zpp = Azp * Pzp, zmm = Azm * Pzm;
    const T zflux = zpp - zmm;
    const T rdr = r * dr;
    T a = r_p * Arp;
    a = a * Prp;
    T b = r_m * Arm;
    b = b * Prm;
    T rterm = a - b;
    rterm = rterm / rdr;
    T v;
    if (k == 0) {                       // z = 0 (written last in the reference: wins on the corners)
      v = (T)2 * zpp;
      v = v / dz;
      v = v + nn(rterm);
    } else if (k == M.nz - 1) {         // z = Z
      T nz_ = -Azm;
      nz_ = nz_ * Pzm;
      v = (T)2 * nz_;
      v = v / dz;
      v = v + nn(rterm);
    } else if (i == 0) {                // r = 0
      v = zflux / dz;
    } else if (i == M.nr - 1) {         // r = R
      T nb = -r_m;
      nb = nb * Arm;
      nb = nb * Prm;
      nb = nb / rdr;
      nb = (T)2 * nb;
      v = zflux / dz;
      v = v + nb;
    } else {
      v = zflux / dz;
      v = v + rterm;
    }
    out[o] = v;
  }
}

template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_rfp_diffusion(DevGeom G, DevEq<T> E, const T* __restrict__ Drr,
                                                             const T* __restrict__ Drz, const T* __restrict__ Dzz,
                                                             const T* __restrict__ f, T* __restrict__ out) {
  const Rz2<T> M{G.n1, G.n2, E.rz + 5 * E.rz_n};
  const T dr = E.grd.h[1], dz = E.grd.h[2], dr2 = E.grd.h2[1], dz2 = E.grd.h2[2];
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < G.ncell; o += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = o / M.nz, k = o - i * M.nz;
    auto F = [&](int di, int dk) { return M.at(f, i, k, di, dk); };
    auto C = [&](int ui, int uk) {  // rfp.py:233-251
      T s = M.at(Drz, i, k, ui, uk) + M.at(Drz, i, k, ui, uk - 1);
      s = s + M.at(Drz, i, k, ui - 1, uk);
      s = s + M.at(Drz, i, k, ui - 1, uk - 1);
      return s / (T)4;
    };
    auto Gd = [&](int pi, int pk, int mi, int mk, T h) { return (F(pi, pk) - F(mi, mk)) / h; };  // rfp.py:221-230
    const T p = f[o];
    T rr_p = (M.at(Drr, i, k, 1, 0) + Drr[o]) * (F(1, 0) - p);
    rr_p = rr_p / dr2;
    T rr_m = (M.at(Drr, i, k, -1, 0) + Drr[o]) * (p - F(-1, 0));
    rr_m = rr_m / dr2;
    T zz_p = (M.at(Dzz, i, k, 0, 1) + Dzz[o]) * (F(0, 1) - p);
    zz_p = zz_p / dz2;
    T zz_m = (M.at(Dzz, i, k, 0, -1) + Dzz[o]) * (p - F(0, -1));
    zz_m = zz_m / dz2;
    const T c_pp = C(1, 1), c_pm = C(1, 0), c_mp = C(0, 1), c_mm = C(0, 0);
    auto mix = [&](T ca, T ga1, T ga2, T cb, T gb1, T gb2) {
      T l = (T)0.25 * ca;
      l = l * (ga1 + ga2);
      T r2 = (T)0.25 * cb;
      r2 = r2 * (gb1 + gb2);
      return l + r2;
    };
    const T rz_r_zp = mix(c_pp, Gd(1, 0, 0, 0, dr), Gd(1, 1, 0, 1, dr), c_mp, Gd(0, 0, -1, 0, dr), Gd(0, 1, -1, 1, dr));
    const T rz_r_zm = mix(c_pm, Gd(1, -1, 0, -1, dr), Gd(1, 0, 0, 0, dr), c_mm, Gd(0, -1, -1, -1, dr), Gd(0, 0, -1, 0, dr));
    const T rz_z_rp = mix(c_pp, Gd(0, 1, 0, 0, dz), Gd(1, 1, 1, 0, dz), c_mp, Gd(0, 0, 0, -1, dz), Gd(1, 0, 1, -1, dz));
    const T rz_z_rm = mix(c_pm, Gd(-1, 1, -1, 0, dz), Gd(0, 1, 0, 0, dz), c_mm, Gd(-1, 0, -1, -1, dz), Gd(0, 0, 0, -1, dz));
    const T r = M.R[i];
    const T r_p = (M.R[pa_wrap(i + 1, M.nr)] + r) / (T)2, r_m = (r + M.R[pa_wrap(i - 1, M.nr)]) / (T)2;
    const T rdr = r * dr;
    T a = r_p * rz_z_rp, b = r_m * rz_z_rm;
    T rad_mixed = a - b;
    rad_mixed = rad_mixed / rdr;
    a = r_p * rr_p;
    b = r_m * rr_m;
    T rad_rr = a - b;
    rad_rr = rad_rr / rdr;
    const T zz = zz_p - zz_m, rzr = rz_r_zp - rz_r_zm;
    T v;
    if (k == 0) {
      T s = zz_p / dz;
      T m = rz_r_zp / dz;
      s = s + m;
      v = (T)2 * s;
      v = v + nn(rad_mixed + rad_rr);
    } else if (k == M.nz - 1) {
      T s = (-zz_m) / dz;
      T m = (-rz_r_zm) / dz;
      s = s + m;
      v = (T)2 * s;
      v = v + nn(rad_mixed + rad_rr);
    } else if (i == 0) {
      v = zz / dz;
      T m = (T)2 * rzr;
      m = m / dz;
      v = v + m;
    } else if (i == M.nr - 1) {
      v = zz / dz;
      T m = rzr / dz;
      v = v + m;
      m = (-r_m) * rz_z_rm;
      m = m / rdr;
      m = (T)2 * m;
      v = v + m;
      m = (-r_m) * rr_m;
      m = m / rdr;
      m = (T)2 * m;
      v = v + m;
    } else {
      v = zz / dz;
      T m = rzr / dz;
      v = v + m;
      v = v + rad_mixed;
      v = v + rad_rr;
    }
    out[o] = v;
  }
}

// rfp.py:262-286
template <typename T>
__device__ __forceinline__ T minmod1(T a, T b) {
  T v = (T)0;
  if (a >= (T)0 && b >= (T)0) v = a < b ? a : b;          // torch.min: NaN handled by the mask tests
  if (a < (T)0 && b < (T)0) v = a > b ? a : b;
  if (a * b <= (T)0) v = (T)0;
  return v;
}

template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_limiter(int64_t n, const T* __restrict__ a, const T* __restrict__ b,
                                                       T* __restrict__ out, int mc) {
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (int64_t)gridDim.x * blockDim.x) {
    const T x = a[o], y = b[o];
    T v = minmod1<T>(x, y);
    if (mc) {
      T s = x + y;
      v = minmod1<T>((T)2 * v, s / (T)2);
    }
    out[o] = v;
  }
}

int need_rz2d(pa_ctx* c, const char* who) {
  if (!c || !c->grid_set) { if (c) pa_set_err(c, "%s before pa_grid_set", who); return PA_E_STATE; }
  if (c->coord != PA_COORD_RZ || c->ndim != 2) {
    pa_set_err(c, "%s is only implemented for the rz coordinate system (rfp.py:22-25)", who);
    return PA_E_ARG;
  }
  return PA_OK;
}

template <typename T>
int div_general_t(pa_ctx* c, const pa_div_spec* s, T* y) {
  pa_term t;
  memset(&t, 0, sizeof(t));
  t.kind = PA_OP_GRAD; t.sign = 1.0;
  DevEq<T> E;
  pa_build_eq<T>(c, 1, &t, E);
  DivSpec<T> S;
  memset(&S, 0, sizeof(S));
  const int sh = 3 - c->ndim;
  const int64_t N[3] = {c->G.n0, c->G.n1, c->G.n2};
  for (int a = 0; a < c->ndim; ++a) {
    if (!s->x[a]) { pa_set_err(c, "pa_div_general: x[%d] is null", a); return PA_E_ARG; }
    if (s->edge && N[a + sh] < 3) { pa_set_err(c, "edge Div needs >= 3 nodes per axis"); return PA_E_ARG; }
    S.x[a + sh] = (const T*)s->x[a];
    S.ui[a + sh] = (const T*)s->u_int[a];
    S.ue[a + sh] = (const T*)s->u_edge[a];
  }
  S.u = (T)s->u;
  S.kind = s->kind;
  S.edge = s->edge;
  hipLaunchKernelGGL(k_div_general<T>, dim3(pa_grid_blocks(c->G.ncell)), dim3(PA_BLOCK), 0, c->stream, c->G, E, S, y);
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

template <typename T>
DevEq<T> plain_eq(pa_ctx* c) {
  pa_term t;
  memset(&t, 0, sizeof(t));
  t.kind = PA_OP_GRAD; t.sign = 1.0;
  DevEq<T> E;
  pa_build_eq<T>(c, 1, &t, E);
  return E;
}

}  // namespace

extern "C" {

int pa_div_general(pa_ctx* c, const pa_div_spec* s, void* y) {
  if (!c || !c->grid_set || !s || !y) { if (c) pa_set_err(c, "pa_div_general: bad arguments"); return c ? PA_E_ARG : PA_E_STATE; }
  if (s->kind != PA_OP_DIV_CENTRAL && s->kind != PA_OP_DIV_UPWIND_COMPAT && s->kind != PA_OP_DIV_UPWIND) {
    pa_set_err(c, "bad div kind %d", s->kind);
    return PA_E_ARG;
  }
  if (s->kind == PA_OP_DIV_CENTRAL)
    for (int f = 0; f < 6; ++f)
      if (c->G.treat[f]) {
        pa_set_err(c, "central Div with neumann/symmetry faces: the reference raises IndexError (fdc.py:583)");
        return PA_E_ARG;
      }
  if (c->ndim == 3 && c->G.n0 != c->G.g0) { pa_set_err(c, "pa_div_general is single-GPU (no slab ghosts)"); return PA_E_ARG; }
  PA_HIP(c, hipSetDevice(c->device));
  return c->dtype == PA_F64 ? div_general_t<double>(c, s, (double*)y) : div_general_t<float>(c, s, (float*)y);
}

int pa_diff_flux(pa_ctx* c, const void* const* D, const void* const* J, void* out) {
  if (!c || !c->grid_set || !D || !J || !out) { if (c) pa_set_err(c, "pa_diff_flux: bad arguments"); return c ? PA_E_ARG : PA_E_STATE; }
  const int nd = c->ndim;
  for (int q = 0; q < nd * nd; ++q) if (!D[q]) { pa_set_err(c, "pa_diff_flux: D[%d] is null", q); return PA_E_ARG; }
  for (int q = 0; q < nd; ++q) if (!J[q]) { pa_set_err(c, "pa_diff_flux: J[%d] is null", q); return PA_E_ARG; }
  PA_HIP(c, hipSetDevice(c->device));
  if (c->dtype == PA_F64) {
    FluxArgs<double> A;
    memset(&A, 0, sizeof(A));
    A.nd = nd;
    for (int q = 0; q < nd * nd; ++q) A.D[q] = (const double*)D[q];
    for (int q = 0; q < nd; ++q) A.J[q] = (const double*)J[q];
    hipLaunchKernelGGL(k_diff_flux<double>, dim3(pa_grid_blocks(c->G.ncell)), dim3(PA_BLOCK), 0, c->stream, c->G,
                       plain_eq<double>(c), A, (double*)out);
  } else {
    FluxArgs<float> A;
    memset(&A, 0, sizeof(A));
    A.nd = nd;
    for (int q = 0; q < nd * nd; ++q) A.D[q] = (const float*)D[q];
    for (int q = 0; q < nd; ++q) A.J[q] = (const float*)J[q];
    hipLaunchKernelGGL(k_diff_flux<float>, dim3(pa_grid_blocks(c->G.ncell)), dim3(PA_BLOCK), 0, c->stream, c->G,
                       plain_eq<float>(c), A, (float*)out);
  }
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

int pa_rfp_friction(pa_ctx* c, const void* Hr, const void* Hz, const void* pdf, void* out) {
  if (int rc = need_rz2d(c, "pa_rfp_friction")) return rc;
  if (!Hr || !Hz || !pdf || !out) { pa_set_err(c, "pa_rfp_friction: null argument"); return PA_E_ARG; }
  PA_HIP(c, hipSetDevice(c->device));
  const dim3 grid(pa_grid_blocks(c->G.ncell)), blk(PA_BLOCK);
  if (c->dtype == PA_F64)
    hipLaunchKernelGGL(k_rfp_friction<double>, grid, blk, 0, c->stream, c->G, plain_eq<double>(c), (const double*)Hr,
                       (const double*)Hz, (const double*)pdf, (double*)out);
  else
    hipLaunchKernelGGL(k_rfp_friction<float>, grid, blk, 0, c->stream, c->G, plain_eq<float>(c), (const float*)Hr,
                       (const float*)Hz, (const float*)pdf, (float*)out);
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

int pa_rfp_diffusion(pa_ctx* c, const void* Drr, const void* Drz, const void* Dzz, const void* pdf, void* out) {
  if (int rc = need_rz2d(c, "pa_rfp_diffusion")) return rc;
  if (!Drr || !Drz || !Dzz || !pdf || !out) { pa_set_err(c, "pa_rfp_diffusion: null argument"); return PA_E_ARG; }
  PA_HIP(c, hipSetDevice(c->device));
  const dim3 grid(pa_grid_blocks(c->G.ncell)), blk(PA_BLOCK);
  if (c->dtype == PA_F64)
    hipLaunchKernelGGL(k_rfp_diffusion<double>, grid, blk, 0, c->stream, c->G, plain_eq<double>(c), (const double*)Drr,
                       (const double*)Drz, (const double*)Dzz, (const double*)pdf, (double*)out);
  else
    hipLaunchKernelGGL(k_rfp_diffusion<float>, grid, blk, 0, c->stream, c->G, plain_eq<float>(c), (const float*)Drr,
                       (const float*)Drz, (const float*)Dzz, (const float*)pdf, (float*)out);
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

int pa_limiter(pa_ctx* c, int which, const void* a, const void* b, void* out, int64_t n) {
  if (!c || !a || !b || !out || n < 0) { if (c) pa_set_err(c, "pa_limiter: bad arguments"); return PA_E_ARG; }
  if (which != 0 && which != 1) { pa_set_err(c, "pa_limiter: 0 = minmod, 1 = monotonized central"); return PA_E_ARG; }
  if (n == 0) return PA_OK;
  PA_HIP(c, hipSetDevice(c->device));
  const dim3 grid(pa_grid_blocks(n)), blk(PA_BLOCK);
  if (c->dtype == PA_F64)
    hipLaunchKernelGGL(k_limiter<double>, grid, blk, 0, c->stream, n, (const double*)a, (const double*)b, (double*)out, which);
  else
    hipLaunchKernelGGL(k_limiter<float>, grid, blk, 0, c->stream, n, (const float*)a, (const float*)b, (float*)out, which);
  PA_HIP(c, hipGetLastError());
  return PA_OK;
}

}  // extern "C"
