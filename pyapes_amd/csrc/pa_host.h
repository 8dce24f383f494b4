// pa_host.h -- host-side context of libpyapes_hip (shared by every translation unit)
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>

#include "../../include/pyapes_hip.h"
#include "pa_device.h"

#define PA_MAX_GRID 2048  // 256 CUs x 8 resident workgroups of 256 threads
#define PA_MAX_PARTIALS 131072  // rows of per-block partial sums (2-D tile grids are one block per tile; 4 MB of scratch)

// device-resident solver scalars: alpha, beta, the stop test and the iteration count never
// leave the GPU inside a solve; the host only polls `done`
struct SolverScalars {
  double rr, rr_old, dAd, alpha, beta, tol, tolerance;
  double rho, omega, rho_next, r0v, ts, tt, r0t;  // bicgstab
  long long itr, max_it;
  int done, err, finished_early;
};

struct HostBC {
  int type = PA_BC_NONE;
  double value = 0.0;
  const void* vals = nullptr;
  double dxf = 0.0;
};

enum {
  SCR_R = 0, SCR_D0, SCR_D1, SCR_PART, SCR_PART2, SCR_SHELL, SCR_R0, SCR_V0, SCR_V1, SCR_S, SCR_TT,
  SCR_GHOST, SCR_SHELL2, SCR_RZ, SCR_RES, PA_NSCRATCH
};

// room behind every scratch allocation: any of the roles r / d / d' of a CG solve fits any block of the placement
// search's pool at its own offset (pa_place.hip)
#define PA_PLACE_ROOM ((size_t)2 << 20)

// ---- online placement search of large CG solves (pa_place.hip) --------------------------------------------------
#define PA_PLACE_MAXSPARE 8
#define PA_PLACE_NEV 8
#define PA_PLACE_NDUR 32
#define PA_PLACE_NX 4
struct PlaceSearch {
  // options "place_minbytes", "place_blocks", "place_budget" (per cent)
  size_t minbytes = (size_t)128 << 20;   // arrays the Infinity Cache holds are left alone
  int blocks = 3;                        // candidates per role = spare allocations held while a pass is on
  double budget = 0.03;                  // what the trials may cost, as a share of the time solved so far
  // pool: blocks no role lives in at the moment
  char* spare[PA_PLACE_MAXSPARE] = {nullptr};
  int spare_epoch[PA_PLACE_MAXSPARE] = {0};   // solve in which the block last carried r / d (zero where the phases skip)
  int nspare = 0;
  size_t bytes = 0, blk = 0;             // array bytes the pool was made for; allocation size of a block
  // the pass: role 0 r, 1 d0, 2 d1 tries candidates 0 .. blocks - 1
  int phase = 0;                         // 0 not begun, 1 searching, 2 over
  int role = 0, cand = 0;
  const void* decided[PA_PLACE_NX] = {nullptr};   // x pointers a pass has been completed for
  int n_decided = 0;
  // the live solve
  int active = 0, st = 0, pause = 0;     // st: 0 measuring the best assignment, 1 trial running, 2 trial to be undone,
                                         //     3 a move of r waits for its phase B (r_move: 1 into the trial block, 2 back)
  int r_move = 0;
  int epoch = 0;
  const void* x = nullptr;
  int64_t it = 0, known = -1, s = 0, clean_from = 1;
  double base = 0.0;                     // best iteration pair under the kept assignment, us
  double dur[PA_PLACE_NDUR] = {0.0};
  double base_par[2] = {0.0, 0.0};       // best single iteration per parity of the direction ping-pong, us
  unsigned char par[PA_PLACE_NDUR] = {0};
  hipEvent_t ev0[PA_PLACE_NEV] = {nullptr}, ev1[PA_PLACE_NEV] = {nullptr};   // start / end of iteration j (ring)
  hipEvent_t evs[2] = {nullptr, nullptr};   // around the copy / memset of the running trial's switch
  int closed = 1, sw_timed = 0;
  // accounts (pa_place_stats)
  int trials = 0, accepted = 0, mallocs = 0;
  double spent_us = 0.0, elapsed_us = 0.0, first_pair = 0.0;
};

struct pa_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  char err[512];
  // grid
  int grid_set = 0, ndim = 0, dtype = PA_F64, esize = 8;
  double dx[3] = {1, 1, 1};
  DevGeom G;
  // BCs (internal face numbering)
  HostBC bc[6];
  int bc_order[6] = {0, 0, 0, 0, 0, 0};
  int nbc = 0;
  // equation
  int eq_set = 0, nterms = 0;
  pa_term terms[PA_MAX_TERMS];
  // scratch
  void* scr[PA_NSCRATCH] = {nullptr};
  void* scr_base[PA_NSCRATCH] = {nullptr};  // what hipMalloc returned (scr[q] = base + q * stagger)
  size_t cap[PA_NSCRATCH] = {0};
  SolverScalars* sc = nullptr;    // device: the CURRENT scalars (one of the two slots of sc_base)
  SolverScalars* sc_alt = nullptr;  // the other slot: a folded phase A writes the next state there, then they swap
  SolverScalars* sc_base = nullptr;
  // folded scalar steps (pa_cg_iterate on one GPU): the single-block kernels between the phases are
  // gone, every block of the NEXT tiled kernel reduces the partial rows itself (pa_cg3d_kernel.h)
  int fold = 1, in_iterate = 0;
  int fold_a_n = 0;                 // rows of d.Ad partials waiting for phase B's prologue
  int fold_b_n = 0, fold_b_nsh = 0; // rows of phase-B (Jacobi: sweep) / shell partials waiting for the next phase A (sweep)
  const double* fold_b_part = nullptr;
  const double* fold_b_shell = nullptr;  // shell rows of that step (null: SCR_PART2)
  // slab iterations with the scalar steps folded (pa_cg_iterate_comm): the per-workgroup partial ROWS are
  // all-reduced instead of their sums -- out of place, rows_send -> rows_recv, so that rows a rank never
  // writes (its grid is smaller than the agreed row count) stay zero -- and the unchanged prologues of the
  // mid kernel / the next phase A sum the all-reduced rows in their fixed order.  Layout of both buffers:
  // [ M_A rows of d.Ad | M_S boundary-shell rows | M_B rows of (r.r, |dx|^2) ], counts = max over ranks.
  int slab_fold = 0;        // row counts agreed (pa_cg_fold_set) for the live solve
  int slab_fold_live = 0;   // inside the folded sequence of pa_cg_iterate_comm
  int fold_rows[3] = {0, 0, 0};
  double* rows_send = nullptr;     // = rows_buf[0]
  double* rows_recv = nullptr;     // = rows_buf[1], or rows_buf[0] when this rank reduces in place
  double* rows_buf[2] = {nullptr, nullptr};
  size_t rows_cap = 0;
  int plan_only = 0;        // launch_cg3d returns its grid size without launching (pa_cg_fold_plan)
  SolverScalars* h_sc = nullptr;  // pinned host mirror
  // pipelined polls of the done flag: the copy of batch b's scalars is waited for after batch b+1 has
  // been enqueued, so the GPU never idles between batches (the over-enqueued batch is no-ops)
  SolverScalars* h_poll[2] = {nullptr, nullptr};
  hipEvent_t ev_poll[2] = {nullptr, nullptr};
  double* sums = nullptr;         // device, PA_NSUM (internal)
  double* ext_sums = nullptr;     // slab: caller-owned sums buffer (all-reduced by the host driver)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_switch = nullptr;   // pa_ctx_set_stream: orders the new stream after what the old one still holds
  // CG state
  int solver_live = 0, cur = 0, bc_static = 0, pending_init_logic = 0, b_blocks = 0;
  int bc_pair = 0;  // per-axis pair kernels (lower + upper face + shell stop-test term in one launch)
  int bc_fused = 0, shell_cur = 0;  // fused BC fill: which half of SCR_SHELL holds x_old on the shell
  void* cg_x = nullptr;
  const void* jac_rhs = nullptr;   // stepwise Jacobi on a slab (pa_jacobi_begin): the right-hand side and omega of the live solve
  double jac_omega = 1.0;
  void* cg_r_out = nullptr;  // placement search: the NEXT phase B writes the new residual here instead of in place
  int64_t cg2d_mincells = 1500000;   // 2-D meshes of at least this many cells run the CG phases on k_cg2d (< 0: never)
  int bcl = 1;               // option "bcl" / PYAPES_HIP_BCL: explicit Euler march without a BC-fill launch per step (pa_sf_kernel.h)
  int place = 1;             // option "place": large CG solves search, while they run, which allocations r / d live in
  PlaceSearch ps;            // ... pa_place.hip
  int pitch = 1;             // option "pitch" / PYAPES_HIP_PITCH: allow that layout (0: odd rows stay on the NARROW kernels)
  // test / A-B switches without an environment variable of their own (pa_ctx_set_option, or PYAPES_HIP_OPTIONS=name=value,...)
  int bc_path = 0;           // "bc_path": bit 0 never the closed-form fill, bit 1 never the per-axis pair kernels, bit 2 closed form at any size
  int bicg_pfold = 1;        // "bicg_pfold": BiCGSTAB's next direction formed by the x / r update (0: the full p / v phase every iteration)
  int bicg_srv = 1;          // "bicg_srv": the tiled s / t phase stores t alone, the x / r update re-forms s = r - alpha v' (0: s stored, as before round 4)
  int bicg_s_stored = 1;     // slab BiCGSTAB: did the s / t step of this iteration store s (else the x / r step re-forms it)
  int jac_alt = 1;           // "jac_alt": consecutive Jacobi sweeps of 3-D meshes march in opposite directions (0: all forwards)
  int jac_dir = 0;           // ... direction of the next sweep
  int rhs_full = 0;          // "rhs_full": pa_rhs_adjust visits the whole mesh instead of the Neumann layers
  int res_cells = 0, res_nt = 0, res_nt_cells = 0, res_spin = -1, res_rzlean = 1;   // "res_*": box plan / threads / spin bound / rz stencil of pa_resident.hip
  int opt_comm = 1;          // "comm" / PYAPES_HIP_COMM: slab driver may use RCCL inside the library (read by pyapes_amd/slab.py)
  int opt_slab_fold = 1;     // "slab_fold" / PYAPES_HIP_SLAB_FOLD: ... with the folded iteration sequence
  int comm_overlap = -1;     // "comm_overlap" / PYAPES_HIP_COMM_OVERLAP: plane exchange on a second communicator + stream (-1: iff N > 1)
  int comm_timeout = 60;     // "comm_timeout" / PYAPES_HIP_COMM_TIMEOUT: seconds a collective of the set-up may take
  int cg_pitch = 0;          // live CG solve keeps r and the direction buffers in the PITCH layout of k_cg3d
  int64_t cg_ps1 = 0;        // ... with this row pitch (cells)
  // Field.VARo (var.save_old() at the top of every solver iteration, linalg.py:110 / 210): when the caller
  // hands a buffer (pa_solver_keep_old) the loops copy the iterate into it before each update -- one extra
  // pass per iteration, paid only on request
  void* x_old_out = nullptr;
  // slab decomposition (P > 1): externally owned exchange buffers
  int slab = 0;
  const void* x_glo = nullptr;   // ghost planes of the field handed to pa_aop / begin
  const void* x_ghi = nullptr;
  void* r_send_lo = nullptr;     // copies of r's first / last owned plane
  void* r_send_hi = nullptr;
  const void* r_recv_lo = nullptr;  // ghost planes of r
  const void* r_recv_hi = nullptr;
  void* d_glo[2] = {nullptr, nullptr};  // ghost planes of the two direction buffers (ctx scratch)
  void* d_ghi[2] = {nullptr, nullptr};
  const void* bc_far_lo0 = nullptr;  // periodic axis-0 fill: x[N-1] (for the lower end rank)
  const void* bc_far_lo1 = nullptr;  //                       x[N-2]
  const void* bc_far_hi0 = nullptr;  //                       x[1] of the lower end rank (for the upper end rank)
  void* v_send_lo = nullptr;         // stepwise BiCGSTAB on a slab (pa_slab_set_v): first / last owned plane of v' out,
  void* v_send_hi = nullptr;
  const void* v_recv_lo = nullptr;   // ... ghost planes of v' in
  const void* v_recv_hi = nullptr;
  void* x_pack_lo1 = nullptr;        // pack destinations of x[1] / x[n0-1] / x[n0-2] (periodic end ranks)
  void* x_pack_hi0 = nullptr;
  void* x_pack_hi1 = nullptr;
  // per-kernel timing of the two dominant CG kernels (pa_profile_set): HIP events on the ctx stream
  int profile = 0;
  hipEvent_t pev[4] = {nullptr, nullptr, nullptr, nullptr};
  double prof_ms[2] = {0.0, 0.0};
  int64_t prof_n[2] = {0, 0};
  // 3-D fast path switch (PYAPES_HIP_FASTPATH=0 disables; tests compare both)
  int fastpath = 1;
  int sf = 1;                    // k_sf for the Div-carrying single-field operations (else k_cg3d's phases)
  int resident = 1;              // small meshes: the whole CG / Jacobi solve in one cooperative launch (pa_resident.hip)
  // 1: hipLaunchCooperativeKernel (the runtime guarantees co-residency).  0: a plain launch of the same grid, which
  // the occupancy query of pa_resident_launch has already shown to fit an idle device; the bounded waits make a
  // grid that was NOT co-resident (device shared with other work) fall back like any other wait that gives up.
  // Exists for collections under rocprofv3: ANY process that has made a cooperative launch dies with SIGSEGV in
  // libhsa-runtime64 inside exit() under rocprofv3 7.2 (profiles/README.md, tools/coop_exit_repro.sh)
  int resident_coop = 1;
  int resident_used = 0;         // workgroups of the last solve's resident launch (0: launch-per-phase loops ran)
  // RCCL communicator owned by the library (pa_comm_*): slab iterations without host work
  void* comm = nullptr;          // ncclComm_t
  // second communicator + stream: the packed plane exchange of an iteration flies beside phase B, the BC
  // fill and the second all-reduce instead of between them (null: exchange on the ctx stream)
  void* comm2 = nullptr;
  hipStream_t xstream = nullptr;
  hipEvent_t ev_b = nullptr, ev_x = nullptr;
  int comm_rank = 0, comm_n = 0;
  pa_exchange plan;
  int plan_set = 0;
  int coord = 0;                 // PA_COORD_*
  const void* rz_tab = nullptr;  // 6 x n_r table built by pa_coord_set (ctx scratch)
};

static inline double* pa_sums(const pa_ctx* c) { return (c->slab && c->ext_sums) ? c->ext_sums : c->sums; }

static inline const int* pa_done_flag(const pa_ctx* c) { return &c->sc->done; }

void pa_set_err(pa_ctx* c, const char* fmt, ...);
// roctx ranges around the solver phases (host-side enqueue ranges; rocprofv3 --marker-trace shows them beside the
// kernel trace).  Off unless PYAPES_HIP_ROCTX=1; libroctx64 is resolved at run time, like librccl.
void pa_range_push(const char* name);
void pa_range_pop();
struct PaRange {
  explicit PaRange(const char* name) { pa_range_push(name); }
  ~PaRange() { pa_range_pop(); }
};
int pa_hip_fail(pa_ctx* c, hipError_t e, const char* what);
int pa_grid_blocks(int64_t work);
int pa_scratch(pa_ctx* c, void** slot, size_t* cap, size_t bytes);
void pa_refresh_geom(pa_ctx* c);
int pa_bc_apply_any(pa_ctx* c, void* x);
int pa_check_eq_applicable(pa_ctx* c);   // pa_ops.hip: Grad inside a solver equation is 1-D only
int pa_cg_slab_mid(pa_ctx* c);                // pa_solver.hip: the step between the phases of a folded slab iteration
int pa_cg_slab_flush(pa_ctx* c);              // close the last iteration of a folded batch (single-block kernel)
int pa_bc_shell_rows(const pa_ctx* c);        // pa_bc.hip: partial rows the BC fill + shell pass of an iteration writes
void pa_profile_stop(pa_ctx* c, int which);   // pa_solver.hip: close the HIP-event bracket of dominant kernel `which`
// pa_place.hip: the online placement search of large CG solves
int pa_place_begin(pa_ctx* c, const void* x, size_t array_bytes);   // pa_cg_begin, before r / d are written
int pa_place_tick(pa_ctx* c);                                        // top of every iteration, before phase A
int pa_place_batch_end(pa_ctx* c);                                   // after the last iteration a call enqueues
void pa_place_r_written(pa_ctx* c);                                  // phase B has written r into c->cg_r_out
int pa_place_prepare_block(pa_ctx* c, void* block);                  // pa_solver.hip: zero what the tiled phases never write
void pa_place_end(pa_ctx* c, int may_free);                          // the solve is over (or dropped); may_free: the stream has been waited for
void pa_place_reset(pa_ctx* c);                                      // the arrays changed: free the pool, forget the pass
void pa_place_destroy(pa_ctx* c);

// ---- BC fill (pa_bc.hip) ---------------------------------------------------------------------------
int pa_shell_blocks(const pa_ctx* c);
int64_t pa_shell_elems(const pa_ctx* c);
bool pa_bc_is_static(const pa_ctx* c);   // every face dirichlet: B(x) is a no-op after the first fill
bool pa_bc_fusable(const pa_ctx* c);     // closed form applies (factory order, small shell)
bool pa_bc_pairable(const pa_ctx* c);    // one launch per axis applies (factory order, both faces present)
template <typename T>
int pa_bc_apply_faces(pa_ctx* c, T* x, bool guarded = false);   // one launch per face, list order
template <typename T>
int pa_bc_apply_auto(pa_ctx* c, T* x, bool guarded);            // fewest launches with the sequential semantics
template <typename T>
int pa_bc_shell_fused(pa_ctx* c, T* x, double* part2, int with_delta, bool guarded, int* nsh, bool standalone);
template <typename T>
int pa_bc_pair_apply(pa_ctx* c, T* x, double* part2, int mode, bool guarded, int* nsh);
template <typename T>
void pa_shell_launch(pa_ctx* c, const T* x, T* shell, double* part2, int with_delta);

#define PA_HIP(c, call)                                              \
  do {                                                               \
    hipError_t e__ = (call);                                         \
    if (e__ != hipSuccess) return pa_hip_fail((c), e__, #call);      \
  } while (0)

template <typename T>
void pa_build_eq(const pa_ctx* c, int nterms, const pa_term* terms, DevEq<T>& E);
template <typename T>
Vec<T> pa_vec_self(const pa_ctx* c, const T* p);

// 3-D fast path (pa_cg3d.hip): return number of partial-sum rows written (> 0) when the
// kernel ran, 0 when the configuration is not covered (caller falls back to the generic
// kernel, which is still HIP), < 0 on error.
template <typename T>
int pa_cg3d_phase_a(pa_ctx* c, const DevEq<T>& E, Vec<T> r, Vec<T> d, T* dnew, double* partials);
template <typename T>
int pa_cg3d_phase_b(pa_ctx* c, const DevEq<T>& E, Vec<T> d, T* x, T* r, double* partials);

// single-field tiled kernels (pa_cg3d.hip); same return convention as the CG phases
// small meshes (pa_resident.hip): the WHOLE solve -- BC fill of the start, first residual, the loop -- in one
// cooperative launch, fields resident in LDS.  Returns the number of workgroups (> 0) when the kernel was launched
// (it then runs to the end of the solve and leaves the scalars in c->sc), 0 when the configuration is not covered
// (the caller runs its launch-per-phase path), < 0 on error.  solver: 0 CG, 1 Jacobi, 2 BiCGSTAB.  x: the start as
// the caller hands it over (BCs not filled), rhs: after pa_rhs_adjust.
template <typename T>
int pa_resident_launch(pa_ctx* c, int solver, T* x, const T* rhs, double tol, int64_t max_it, double omega);

template <typename T>
int pa_tile3d_aop(pa_ctx* c, const DevEq<T>& E, Vec<T> x, T* y, int interior_only);
template <typename T>
int pa_tile3d_grad(pa_ctx* c, Vec<T> x, T* y, int nd);
template <typename T>
int pa_tile3d_euler(pa_ctx* c, Vec<T> phi, T* out, int kind, double u, const void* u_field, double nu, double dt,
                    int bcl = 0);   // bcl: "BC on load" (pa_sf_kernel.h); 0 is returned when that form does not apply
template <typename T>
int pa_tile3d_jacobi(pa_ctx* c, const DevEq<T>& E, Vec<T> x, const T* rhs, T* xnew, double omega, double* partials);
template <typename T>
int pa_tile3d_bicg_pv(pa_ctx* c, const DevEq<T>& E, Vec<T> r, Vec<T> p, Vec<T> v, const T* r0, T* pnew, T* vnew,
                      double* partials);
template <typename T>
int pa_tile3d_bicg_v(pa_ctx* c, const DevEq<T>& E, Vec<T> p, const T* r0, T* vnew, double* partials);
template <typename T>
int pa_tile3d_bicg_st(pa_ctx* c, const DevEq<T>& E, Vec<T> r, Vec<T> v, const T* r0, T* s_out, T* t_out,
                      double* partials);
