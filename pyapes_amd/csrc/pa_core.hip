// pa_core.hip -- host infrastructure of libpyapes_hip: context, scratch, grid / coordinate system / BC list /
// equation state behind the C ABI declared in include/pyapes_hip.h.  gfx950 only.
// The kernels live beside the host code that launches them: pa_bc.hip (BC fill), pa_ops.hip (generic
// operators, explicit entry points), pa_solver.hip (generic solver kernels, CG / Jacobi / BiCGSTAB
// drivers), pa_cg3d*.hip (the tiled marching kernel), pa_rfp.hip, pa_comm.hip.
#include "pa_host.h"

#include <dlfcn.h>
#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <new>

// ---- axisymmetric meshes: the r-dependent coefficient rows, once per mesh -------------------------
// Row q of the 6 x n_r table (literal expressions of the reference, evaluated in the grid dtype):
//   0  (1 + s) / dr^2, s = nan_to_num(dr / (2 r))       Laplacian Ap along r   (tools.py:86-99)
//   1  (1 - s) / dr^2                                    Laplacian Am along r   (tools.py:101-106)
//   2  (2/3 + nan_to_num(2/3 dr / r)) / dr^2            neumann / symmetry row (fdc.py:395-417)
//   3  2/3 - nan_to_num(1/3 dr / r)                      rhs adjustment factor  (fdc.py:440-453)
//   4  nan_to_num(2 dr / r)                              Ac of Div along r      (tools.py:64-78)
//   5  r
template <typename T>
__global__ void __launch_bounds__(PA_BLOCK) k_rz_tables(int64_t nr, const T* __restrict__ r_nodes, T dr,
                                                         T* __restrict__ tab) {
  auto nn = [](T v) -> T { return (isnan(v) || isinf(v)) ? (T)0 : v; };
  const T h2 = dr * dr;                       // dx[0] ** 2
  const T c23 = (T)(2.0 / 3.0), c13 = (T)(1.0 / 3.0);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nr; i += (int64_t)gridDim.x * blockDim.x) {
    const T r = r_nodes[i];
    T t = (T)2 * r;
    const T s = nn(dr / t);
    T ap = (T)1 + s;
    T am = (T)1 - s;
    tab[i] = ap / h2;
    tab[nr + i] = am / h2;
    t = c23 * dr;
    T a = nn(t / r);
    a = c23 + a;
    tab[2 * nr + i] = a / h2;
    t = c13 * dr;
    a = nn(t / r);
    tab[3 * nr + i] = c23 - a;
    t = (T)2 * dr;
    tab[4 * nr + i] = nn(t / r);
    tab[5 * nr + i] = r;
  }
}

// ============================================================================
//  host side
// ============================================================================

static thread_local char g_create_err[512] = "";



void pa_set_err(pa_ctx* c, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  if (c) vsnprintf(c->err, sizeof(c->err), fmt, ap);
  else vsnprintf(g_create_err, sizeof(g_create_err), fmt, ap);
  va_end(ap);
}

namespace {
struct Roctx {
  int want = -1;   // -1: ask PYAPES_HIP_ROCTX; 0 / 1: option "roctx"
  int state = 0;   // 0 untried, 1 on, -1 off
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
};
Roctx g_roctx;
bool roctx_on() {
  Roctx& R = g_roctx;
  if (R.state == 0) {
    R.state = -1;
    const char* e = getenv("PYAPES_HIP_ROCTX");
    if (R.want >= 0 ? R.want != 0 : (e && atoi(e) != 0)) {
      void* h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
      if (h) {
        R.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        R.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (R.push && R.pop) R.state = 1;
      }
    }
  }
  return R.state == 1;
}
}  // namespace

void pa_range_push(const char* name) { if (roctx_on()) (void)g_roctx.push(name); }
void pa_range_pop() { if (roctx_on()) (void)g_roctx.pop(); }

int pa_hip_fail(pa_ctx* c, hipError_t e, const char* what) {
  pa_set_err(c, "HIP error in %s: %s", what, hipGetErrorString(e));
  return PA_E_HIP;
}

int pa_grid_blocks(int64_t work) {
  int64_t b = (work + PA_BLOCK - 1) / PA_BLOCK;
  if (b < 1) b = 1;
  if (b > PA_MAX_GRID) b = PA_MAX_GRID;
  return (int)b;
}

int pa_scratch(pa_ctx* c, void** slot, size_t* cap, size_t bytes) {
  if (*cap >= bytes && *slot) return PA_OK;
  const int q = (int)(slot - c->scr);  // every caller passes &c->scr[id]
  // (r / d / d' change size: the pool of the placement search was made for the old one)
  if (*slot && (q == SCR_R || q == SCR_D0 || q == SCR_D1)) pa_place_reset(c);
  if (*slot) { (void)hipFree(c->scr_base[q]); c->scr_base[q] = nullptr; *slot = nullptr; *cap = 0; }
  if (bytes == 0) return PA_OK;
  // hipMalloc returns 2 MB-aligned blocks; fields that are streamed in lockstep (x, r, d, d') would then sit at the
  // same offset of their pages at every moment, and whether the memory channels they fall on coincide is left to
  // where the driver happened to put the pages: on one box of the pool CG phase B at 512^3 fp64 took 929 us with
  // every array at offset 0 and 853 us with the direction buffers 68.25 / 136.5 KiB in (phase A 571 -> 583, the
  // iteration 1.557 -> 1.492 ms); on two other boxes the offsets change nothing (1.512-1.520 ms in every setting) --
  // the box-to-box spread of the headline kernel is page placement, not clocks.  Slot q starts
  // (q + 1) * stagger bytes in: 68 KiB + 256 B.
  const long stagger = 69888, sbase = 1;   // (slot 0, r, one step in as well: x, the caller's array, is the one at offset 0)
  const size_t off = (size_t)(q + sbase) * (size_t)stagger;
  void* base = nullptr;
  // (PA_PLACE_ROOM more: every role of a CG solve fits every block of the placement search's pool, pa_place.hip)
  {
    static const bool dbg = getenv("PYAPES_HIP_DEBUG") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    PA_HIP(c, hipMalloc(&base, bytes + off + PA_PLACE_ROOM));
    if (dbg && bytes >= ((size_t)32 << 20))
      fprintf(stderr, "[pyapes_hip] scratch slot %d: hipMalloc of %.0f MiB: %.0f us\n", q, bytes / 1048576.0,
              std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
  c->scr_base[q] = base;
  *slot = (char*)base + off;
  *cap = bytes;
  return PA_OK;
}

template <typename T>
static void fill_coefs(const pa_ctx* c, DevEq<T>& E) {
  for (int a = 0; a < 3; ++a) {
    T h = (T)c->dx[a];
    T h2 = h * h;
    E.lap.inv[a] = (T)1 / h2;
    E.lap.m2inv[a] = (T)-2 / h2;
    E.lap.c23[a] = (T)(2.0 / 3.0) / h2;
    T th = (T)2 * h;
    T third = (T)(1.0 / 3.0);
    E.grd.h[a] = h;
    E.grd.ih[a] = (T)1 / h;
    E.grd.h2[a] = th;
    E.grd.g[a] = (T)1 / th;
    E.grd.mg[a] = (T)-1 / th;
    T v = (T)1 + third;
    E.grd.lo_p[a] = v / th;
    v = (T)0 - third;
    E.grd.lo_c[a] = v / th;
    v = (T)0 + third;
    E.grd.hi_c[a] = v / th;
    v = (T)-1 - third;
    E.grd.hi_m[a] = v / th;
  }
}

template <typename T>
void pa_build_eq(const pa_ctx* c, int nterms, const pa_term* terms, DevEq<T>& E) {
  memset(&E, 0, sizeof(E));
  E.nterms = nterms;
  for (int q = 0; q < nterms; ++q) {
    E.t[q].kind = terms[q].kind;
    E.t[q].has_coeff = terms[q].has_coeff;
    E.t[q].sign = (T)terms[q].sign;
    E.t[q].coeff = (T)terms[q].coeff;
    E.t[q].coeff_f = (const T*)terms[q].coeff_field;
    E.t[q].u = (T)terms[q].u;
    E.t[q].u_f = (const T*)terms[q].u_field;
  }
  fill_coefs<T>(c, E);
  E.rz = c->coord == PA_COORD_RZ ? (const T*)c->rz_tab : nullptr;
  E.rz_n = c->G.n1;
}
template void pa_build_eq<float>(const pa_ctx*, int, const pa_term*, DevEq<float>&);
template void pa_build_eq<double>(const pa_ctx*, int, const pa_term*, DevEq<double>&);

// interior set, BC flags (mesh/tools.py:7-20, bcs.py:158-163)
void pa_refresh_geom(pa_ctx* c) {
  DevGeom& G = c->G;
  for (int a = 0; a < 3; ++a) {
    int64_t N = a == 0 ? G.g0 : (a == 1 ? G.n1 : G.n2);
    if (!G.act[a]) { G.slo[a] = 0; G.shi[a] = 0; continue; }
    G.slo[a] = 1;
    G.shi[a] = N - 2;
    int lo_pos = -1, hi_pos = -1;
    for (int w = 0; w < c->nbc; ++w) {
      if (c->bc_order[w] == 2 * a) lo_pos = w;
      if (c->bc_order[w] == 2 * a + 1) hi_pos = w;
    }
    G.hi_last[a] = hi_pos > lo_pos;
  }
  for (int f = 0; f < 6; ++f) {
    int t = c->bc[f].type;
    G.bct[f] = t;
    G.treat[f] = (t == PA_BC_NEUMANN || t == PA_BC_SYMMETRY);
    if (t == PA_BC_PERIODIC) {
      int a = f >> 1;
      int64_t N = a == 0 ? G.g0 : (a == 1 ? G.n1 : G.n2);
      if ((f & 1) == 0) G.slo[a] = 0; else G.shi[a] = N - 1;
    }
  }
}

template <typename T>
Vec<T> pa_vec_self(const pa_ctx* c, const T* p) {
  // P = 1 (or no exchange needed): ghost planes are the field's own wrap-around planes
  Vec<T> v;
  v.p = p;
  v.glo = p + (c->G.n0 - 1) * c->G.s0;
  v.ghi = p;
  return v;
}
template Vec<float> pa_vec_self<float>(const pa_ctx*, const float*);
template Vec<double> pa_vec_self<double>(const pa_ctx*, const double*);

// ---------------------------------------------------------------------------------------
extern "C" {

const char* pa_version(void) { return "pyapes_hip 0.2 (gfx950)"; }

const char* pa_last_error(const pa_ctx* c) { return c ? c->err : g_create_err; }

int pa_ctx_create(int device, void* hip_stream, pa_ctx** out) {
  if (!out) { pa_set_err(nullptr, "pa_ctx_create: out is NULL"); return PA_E_ARG; }
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    pa_set_err(nullptr, "pa_ctx_create: no HIP device visible (%s)", hipGetErrorString(e));
    return PA_E_HIP;
  }
  if (device < 0 || device >= ndev) { pa_set_err(nullptr, "pa_ctx_create: bad device %d", device); return PA_E_ARG; }
  e = hipSetDevice(device);
  if (e != hipSuccess) { pa_set_err(nullptr, "hipSetDevice: %s", hipGetErrorString(e)); return PA_E_HIP; }
  pa_ctx* c = new (std::nothrow) pa_ctx();
  if (!c) { pa_set_err(nullptr, "out of host memory"); return PA_E_ARG; }
  c->device = device;
  c->stream = (hipStream_t)hip_stream;
  c->err[0] = 0;
  // Defaults of the options from the environment (the complete list: README.md "Switches"): nine named variables for
  // the switches people flip, PYAPES_HIP_OPTIONS="name=value,..." for everything pa_ctx_set_option knows.
  static const struct { const char* env; const char* opt; } named[] = {
      {"PYAPES_HIP_FASTPATH", "fastpath"}, {"PYAPES_HIP_SF", "sf"}, {"PYAPES_HIP_FOLD", "fold"}, {"PYAPES_HIP_RESIDENT", "resident"},
      {"PYAPES_HIP_PITCH", "pitch"}, {"PYAPES_HIP_BCL", "bcl"}, {"PYAPES_HIP_COMM", "comm"}, {"PYAPES_HIP_SLAB_FOLD", "slab_fold"},
      {"PYAPES_HIP_COMM_OVERLAP", "comm_overlap"}, {"PYAPES_HIP_COMM_TIMEOUT", "comm_timeout"}};
  int orc = PA_OK;
  for (const auto& nv : named)
    if (const char* v = getenv(nv.env)) orc = orc ? orc : pa_ctx_set_option(c, nv.opt, atoi(v));
  if (const char* pl = getenv("PYAPES_HIP_PLACE")) {   // 0 off, 1 on (default), 2 stress: every CG solve of any size, no budget
    c->place = atoi(pl) != 0;
    if (atoi(pl) == 2) { c->ps.minbytes = 0; c->ps.budget = 1e9; }
  }
  if (const char* all = getenv("PYAPES_HIP_OPTIONS")) {
    char buf[1024];
    strncpy(buf, all, sizeof(buf) - 1);
    buf[sizeof(buf) - 1] = 0;
    for (char* tok = strtok(buf, ","); tok && !orc; tok = strtok(nullptr, ",")) {
      char* eq = strchr(tok, '=');
      if (!eq) { pa_set_err(c, "PYAPES_HIP_OPTIONS: '%s' is not name=value", tok); orc = PA_E_ARG; break; }
      *eq = 0;
      orc = pa_ctx_set_option(c, tok, atoi(eq + 1));
    }
  }
  if (orc) {
    pa_set_err(nullptr, "pa_ctx_create: %s", c->err);
    delete c;
    return orc;
  }
  if (hipMalloc((void**)&c->sc_base, 2 * sizeof(SolverScalars)) != hipSuccess ||
      hipMalloc((void**)&c->sums, PA_NSUM * sizeof(double)) != hipSuccess ||
      hipHostMalloc((void**)&c->h_sc, sizeof(SolverScalars)) != hipSuccess ||
      hipHostMalloc((void**)&c->h_poll[0], sizeof(SolverScalars)) != hipSuccess ||
      hipHostMalloc((void**)&c->h_poll[1], sizeof(SolverScalars)) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_poll[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_poll[1], hipEventDisableTiming) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    pa_set_err(nullptr, "pa_ctx_create: allocation failed");
    delete c;
    return PA_E_HIP;
  }
  c->sc = c->sc_base;
  c->sc_alt = c->sc_base + 1;
  (void)hipMemsetAsync(c->sc_base, 0, 2 * sizeof(SolverScalars), c->stream);
  (void)hipMemsetAsync(c->sums, 0, PA_NSUM * sizeof(double), c->stream);
  *out = c;
  return PA_OK;
}

int pa_ctx_set_stream(pa_ctx* c, void* hip_stream) {
  if (!c) return PA_E_ARG;
  if (c->solver_live && (hipStream_t)hip_stream != c->stream) {
    pa_set_err(c, "pa_ctx_set_stream during a stepwise solve");
    return PA_E_STATE;
  }
  hipStream_t next = (hipStream_t)hip_stream;
  if (next != c->stream) {
    // work still queued on the old stream(s) -- asynchronous explicit operators, uploads of BC values, the
    // create-time memsets of the scalars and partial rows, the exchange stream of a slab -- writes ctx-owned
    // state the kernels on the new stream read: order the new stream after it
    PA_HIP(c, hipSetDevice(c->device));
    if (!c->ev_switch) PA_HIP(c, hipEventCreateWithFlags(&c->ev_switch, hipEventDisableTiming));
    // (the OLD stream may be gone by now -- an external, short-lived stream the caller handed in and destroyed: the
    // record fails then.  The switch must still complete: what that stream held has either run or died with it, and a
    // device-wide wait is the conservative order.)
    auto order_after = [&](hipStream_t from) {
      if (hipEventRecord(c->ev_switch, from) == hipSuccess && hipStreamWaitEvent(next, c->ev_switch, 0) == hipSuccess) return;
      (void)hipGetLastError();
      (void)hipDeviceSynchronize();
      (void)hipGetLastError();
    };
    order_after(c->stream);
    if (c->xstream) order_after(c->xstream);
  }
  c->stream = next;
  return PA_OK;
}

int pa_ctx_set_option(pa_ctx* c, const char* name, int value) {
  if (!c || !name) return PA_E_ARG;
  if (c->solver_live) { pa_set_err(c, "pa_ctx_set_option during a solve"); return PA_E_STATE; }
  if (!strcmp(name, "fastpath")) c->fastpath = value != 0;      // tiled kernels (else the generic ones)
  else if (!strcmp(name, "sf")) c->sf = value != 0;             // k_sf (else k_cg3d's single-field phases)
  else if (!strcmp(name, "fold")) c->fold = value != 0;         // scalar steps in the next kernel's prologue
  else if (!strcmp(name, "resident")) c->resident = value != 0; // small meshes: one cooperative launch per solve
  else if (!strcmp(name, "cg2d_mincells")) c->cg2d_mincells = value;   // k_cg2d from this many cells on (< 0: never)
  else if (!strcmp(name, "bcl")) c->bcl = value != 0;         // Euler march: face values formed on load, one fill at the end
  else if (!strcmp(name, "pitch")) c->pitch = value != 0;     // odd row lengths: pitched r / d buffers in the CG phases
  else if (!strcmp(name, "place")) c->place = value != 0;     // large CG solves: online search for the allocations of r / d (pa_place.hip)
  else if (!strcmp(name, "place_minbytes")) c->ps.minbytes = value < 0 ? 0 : (size_t)value;   // ... for arrays of at least this size (tests: 0)
  else if (!strcmp(name, "place_blocks")) c->ps.blocks = value < 0 ? 0 : (value > PA_PLACE_MAXSPARE ? PA_PLACE_MAXSPARE : value);
  else if (!strcmp(name, "place_budget")) c->ps.budget = 0.01 * (value < 0 ? 0 : value);      // per cent of the time solved so far
  // 0: plain launch of the same grid.  PROFILING ONLY (rocprofv3 7.2 kills any process that made a cooperative launch,
  // profiles/README.md): co-residency is then only what an occupancy query on an idle device promised; on a device
  // shared with other work the bounded waits give up and the launch-per-phase loop runs (slow, never wrong)
  else if (!strcmp(name, "resident_coop")) c->resident_coop = value != 0;
  else if (!strcmp(name, "bc_path")) c->bc_path = value & 7;
  else if (!strcmp(name, "bicg_pfold")) c->bicg_pfold = value != 0;
  else if (!strcmp(name, "bicg_srv")) c->bicg_srv = value != 0;
  else if (!strcmp(name, "jac_alt")) c->jac_alt = value != 0;
  else if (!strcmp(name, "rhs_full")) c->rhs_full = value != 0;
  else if (!strcmp(name, "res_cells")) c->res_cells = value;
  else if (!strcmp(name, "res_nt")) c->res_nt = value;
  else if (!strcmp(name, "res_nt_cells")) c->res_nt_cells = value;
  else if (!strcmp(name, "res_spin")) c->res_spin = value;
  else if (!strcmp(name, "res_rzlean")) c->res_rzlean = value != 0;
  else if (!strcmp(name, "comm")) c->opt_comm = value != 0;
  else if (!strcmp(name, "slab_fold")) c->opt_slab_fold = value != 0;
  else if (!strcmp(name, "comm_overlap")) c->comm_overlap = value < 0 ? -1 : (value != 0);
  else if (!strcmp(name, "comm_timeout")) c->comm_timeout = value > 0 ? value : 60;
  else if (!strcmp(name, "roctx")) { g_roctx.want = value != 0; g_roctx.state = 0; }   // (process-wide, like the variable)
  else { pa_set_err(c, "pa_ctx_set_option: unknown option '%s'", name); return PA_E_ARG; }
  return PA_OK;
}

int pa_ctx_get_option(const pa_ctx* c, const char* name, int* value) {
  if (!c || !name || !value) return PA_E_ARG;
  const struct { const char* n; int v; } tab[] = {
      {"fastpath", c->fastpath}, {"sf", c->sf}, {"fold", c->fold}, {"resident", c->resident}, {"bcl", c->bcl}, {"pitch", c->pitch},
      {"place", c->place}, {"resident_coop", c->resident_coop}, {"bc_path", c->bc_path}, {"bicg_pfold", c->bicg_pfold}, {"bicg_srv", c->bicg_srv}, {"jac_alt", c->jac_alt},
      {"rhs_full", c->rhs_full}, {"comm", c->opt_comm}, {"slab_fold", c->opt_slab_fold}, {"comm_overlap", c->comm_overlap},
      {"comm_timeout", c->comm_timeout}, {"place_blocks", c->ps.blocks}};
  for (const auto& t : tab)
    if (!strcmp(name, t.n)) { *value = t.v; return PA_OK; }
  return PA_E_ARG;
}

int pa_ctx_destroy(pa_ctx* c) {
  if (!c) return PA_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)pa_comm_destroy(c);
  pa_place_destroy(c);
  for (int q = 0; q < PA_NSCRATCH; ++q)
    if (c->scr_base[q]) (void)hipFree(c->scr_base[q]);
  if (c->sc_base) (void)hipFree(c->sc_base);
  if (c->sums) (void)hipFree(c->sums);
  if (c->rows_buf[0]) (void)hipFree(c->rows_buf[0]);
  if (c->rows_buf[1]) (void)hipFree(c->rows_buf[1]);
  if (c->h_sc) (void)hipHostFree(c->h_sc);
  for (int q = 0; q < 2; ++q) {
    if (c->h_poll[q]) (void)hipHostFree(c->h_poll[q]);
    if (c->ev_poll[q]) (void)hipEventDestroy(c->ev_poll[q]);
  }
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->ev_switch) (void)hipEventDestroy(c->ev_switch);
  for (int q = 0; q < 4; ++q)
    if (c->pev[q]) (void)hipEventDestroy(c->pev[q]);
  delete c;
  return PA_OK;
}

int pa_grid_set(pa_ctx* c, int ndim, const int64_t* n, const double* dx, int dtype, int64_t i_off,
                int64_t n0_global) {
  if (!c) return PA_E_ARG;
  if (ndim < 1 || ndim > 3) { pa_set_err(c, "pa_grid_set: ndim must be 1..3"); return PA_E_ARG; }
  if (dtype != PA_F32 && dtype != PA_F64) { pa_set_err(c, "pa_grid_set: bad dtype"); return PA_E_ARG; }
  for (int a = 0; a < ndim; ++a)
    if (n[a] < 3 && !(a == 0 && n0_global >= 3 && n[a] >= 1)) {
      pa_set_err(c, "pa_grid_set: every axis needs >= 3 nodes (linalg.py:43-45)");
      return PA_E_ARG;
    }
  if (ndim < 3 && (i_off != 0 || n0_global != n[0])) {
    pa_set_err(c, "pa_grid_set: slab decomposition is for 3-D meshes only (1-D/2-D: replicas)");
    return PA_E_ARG;
  }
  if (i_off < 0 || i_off + n[0] > n0_global) { pa_set_err(c, "pa_grid_set: slab outside the global grid"); return PA_E_ARG; }
  DevGeom& G = c->G;
  memset(&G, 0, sizeof(G));
  int64_t ext[3] = {1, 1, 1};
  double h[3] = {1.0, 1.0, 1.0};
  const int sh = 3 - ndim;
  for (int a = 0; a < ndim; ++a) { ext[a + sh] = n[a]; h[a + sh] = dx[a]; G.act[a + sh] = 1; }
  G.n0 = ext[0]; G.n1 = ext[1]; G.n2 = ext[2];
  G.s1 = G.n2; G.s0 = G.n1 * G.n2;
  G.ncell = G.n0 * G.n1 * G.n2;
  G.off0 = ndim == 3 ? i_off : 0;
  G.g0 = ndim == 3 ? n0_global : 1;
  for (int a = 0; a < 3; ++a) c->dx[a] = h[a];
  c->ndim = ndim;
  c->dtype = dtype;
  c->esize = dtype == PA_F64 ? 8 : 4;
  c->grid_set = 1;
  c->eq_set = 0;
  c->solver_live = 0;
  for (int f = 0; f < 6; ++f) c->bc[f] = HostBC();
  c->nbc = 0;
  c->coord = PA_COORD_XYZ;
  c->rz_tab = nullptr;
  pa_refresh_geom(c);
  pa_place_reset(c);   // (another grid: what a placement search learnt about the old arrays is void)
  return PA_OK;
}

int pa_coord_set(pa_ctx* c, int coord_sys, const void* r_nodes) {
  if (!c || !c->grid_set) { if (c) pa_set_err(c, "pa_coord_set before pa_grid_set"); return PA_E_STATE; }
  if (coord_sys == PA_COORD_XYZ) { c->coord = PA_COORD_XYZ; c->rz_tab = nullptr; return PA_OK; }
  if (coord_sys != PA_COORD_RZ) { pa_set_err(c, "pa_coord_set: unknown coordinate system %d", coord_sys); return PA_E_ARG; }
  if (c->ndim != 2) { pa_set_err(c, "pa_coord_set: rz coordinate system only accepts 2-D grids (_mesh.py:48-49)"); return PA_E_ARG; }
  if (!r_nodes) { pa_set_err(c, "pa_coord_set: rz needs the r coordinates of the nodes"); return PA_E_ARG; }
  PA_HIP(c, hipSetDevice(c->device));
  const int64_t nr = c->G.n1;
  int rc = pa_scratch(c, &c->scr[SCR_RZ], &c->cap[SCR_RZ], (size_t)6 * nr * c->esize);
  if (rc) return rc;
  if (c->dtype == PA_F64)
    hipLaunchKernelGGL(k_rz_tables<double>, dim3(pa_grid_blocks(nr)), dim3(PA_BLOCK), 0, c->stream, nr,
                       (const double*)r_nodes, (double)c->dx[1], (double*)c->scr[SCR_RZ]);
  else
    hipLaunchKernelGGL(k_rz_tables<float>, dim3(pa_grid_blocks(nr)), dim3(PA_BLOCK), 0, c->stream, nr,
                       (const float*)r_nodes, (float)c->dx[1], (float*)c->scr[SCR_RZ]);
  PA_HIP(c, hipGetLastError());
  c->coord = PA_COORD_RZ;
  c->rz_tab = c->scr[SCR_RZ];
  c->solver_live = 0;
  return PA_OK;
}

// The BC list and the equation are read again by every phase of a live stepwise solve (pa_cg_begin ...
// pa_cg_end): changing them in between would silently change the fill and the stencil of the running solve.
static int pa_refuse_live(pa_ctx* c, const char* what) {
  if (!c->solver_live) return PA_OK;
  pa_set_err(c, "%s during a stepwise solve (pa_cg_begin ... pa_cg_end); end or abort it first", what);
  return PA_E_STATE;
}

int pa_bc_clear(pa_ctx* c) {
  if (!c || !c->grid_set) return PA_E_STATE;
  if (int rl = pa_refuse_live(c, "pa_bc_clear")) return rl;
  for (int f = 0; f < 6; ++f) c->bc[f] = HostBC();
  c->nbc = 0;
  pa_refresh_geom(c);
  return PA_OK;
}

int pa_bc_set(pa_ctx* c, int face, int order_pos, int type, double value, const void* face_vals, double dxf) {
  if (!c || !c->grid_set) { if (c) pa_set_err(c, "pa_bc_set before pa_grid_set"); return PA_E_STATE; }
  if (int rl = pa_refuse_live(c, "pa_bc_set")) return rl;
  if (face < 0 || face >= 2 * c->ndim) { pa_set_err(c, "pa_bc_set: face %d outside a %d-D mesh", face, c->ndim); return PA_E_ARG; }
  if (order_pos < 0 || order_pos >= 6) { pa_set_err(c, "pa_bc_set: bad order_pos"); return PA_E_ARG; }
  if (type < PA_BC_NONE || type > PA_BC_PERIODIC) { pa_set_err(c, "pa_bc_set: bad type"); return PA_E_ARG; }
  int fi = face + 2 * (3 - c->ndim);
  HostBC& b = c->bc[fi];
  b.type = type; b.value = value; b.vals = face_vals; b.dxf = dxf;
  c->bc_order[order_pos] = fi;
  if (order_pos + 1 > c->nbc) c->nbc = order_pos + 1;
  pa_refresh_geom(c);
  return PA_OK;
}

int pa_eq_set(pa_ctx* c, int nterms, const pa_term* terms) {
  if (!c || !c->grid_set) { if (c) pa_set_err(c, "pa_eq_set before pa_grid_set"); return PA_E_STATE; }
  if (int rl = pa_refuse_live(c, "pa_eq_set")) return rl;
  if (nterms < 1 || nterms > PA_MAX_TERMS) { pa_set_err(c, "pa_eq_set: 1..%d terms", PA_MAX_TERMS); return PA_E_ARG; }
  for (int q = 0; q < nterms; ++q) {
    int k = terms[q].kind;
    if (k < PA_OP_LAPLACIAN || k > PA_OP_DIV_UPWIND) { pa_set_err(c, "pa_eq_set: bad kind"); return PA_E_ARG; }
    if (k == PA_OP_DIV_CENTRAL)
      for (int f = 0; f < 6; ++f)
        if (c->G.treat[f]) {
          pa_set_err(c, "central Div with neumann/symmetry faces: the reference raises IndexError (fdc.py:583)");
          return PA_E_ARG;
        }
    if (k == PA_OP_DIV_CENTRAL && terms[q].u_field && c->G.n0 != c->G.g0 && c->ndim == 3) {
      pa_set_err(c, "tensor-u central Div is single-GPU only");
      return PA_E_ARG;
    }
  }
  c->nterms = nterms;
  for (int q = 0; q < nterms; ++q) c->terms[q] = terms[q];
  c->eq_set = 1;
  return PA_OK;
}

}  // extern "C"

extern "C" {

int pa_slab_set(pa_ctx* c, const pa_slab* s) {
  if (!c || !c->grid_set) { if (c) pa_set_err(c, "pa_slab_set before pa_grid_set"); return PA_E_STATE; }
  if (c->solver_live) { pa_set_err(c, "pa_slab_set during a solve"); return PA_E_STATE; }
  if (!s) {
    c->slab = 0;
    c->ext_sums = nullptr;
    c->x_glo = c->x_ghi = nullptr;
    c->r_send_lo = c->r_send_hi = nullptr;
    c->r_recv_lo = c->r_recv_hi = nullptr;
    c->bc_far_lo0 = c->bc_far_lo1 = c->bc_far_hi0 = nullptr;
    c->x_pack_lo1 = c->x_pack_hi0 = c->x_pack_hi1 = nullptr;
    return PA_OK;
  }
  if (c->ndim != 3) { pa_set_err(c, "pa_slab_set: slabs are for 3-D meshes"); return PA_E_ARG; }
  if (!s->sums) { pa_set_err(c, "pa_slab_set: sums buffer is required"); return PA_E_ARG; }
  c->slab = 1;
  c->ext_sums = (double*)s->sums;
  c->r_send_lo = s->r_send_lo; c->r_send_hi = s->r_send_hi;
  c->r_recv_lo = s->r_recv_lo; c->r_recv_hi = s->r_recv_hi;
  c->x_glo = s->x_ghost_lo; c->x_ghi = s->x_ghost_hi;
  c->bc_far_lo0 = s->bc_far_lo0; c->bc_far_lo1 = s->bc_far_lo1; c->bc_far_hi0 = s->bc_far_hi0;
  c->x_pack_lo1 = s->x_pack_lo1; c->x_pack_hi0 = s->x_pack_hi0; c->x_pack_hi1 = s->x_pack_hi1;
  return PA_OK;
}

}  // extern "C"

