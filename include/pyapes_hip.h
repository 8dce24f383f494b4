/*
 * pyapes_hip.h -- C ABI of libpyapes_hip.so, the MI355X (gfx950) FDM stencil +
 * iterative-solve core that drops in under pyapes/solver and pyapes/backend.
 *
 * The reference (kyoungseoun-chung/pyapes v0.2.13) is pure Python/torch and has
 * no FFI; the seams this library replaces are the Python call sites listed per
 * entry point below (paths relative to the reference repo).  The host side that
 * binds this ABI with ctypes is pyapes_amd/hip/lib.py; INTEGRATION.md shows the
 * stub a pyapes maintainer would add.
 *
 * Conventions
 *   - Every pointer named x/y/rhs/phi/... is a DEVICE pointer to a C-contiguous
 *     scalar field (n0[,n1[,n2]]) of the ctx dtype (float or double).  The caller
 *     (PyTorch-ROCm tensors, tensor.data_ptr()) owns all of them; the library
 *     never allocates user-visible memory, only its own scratch.
 *   - Faces are numbered 0..5 = xl,xu,yl,yu,zl,zu (pyapes/geometry/basis.py:16).
 *   - One ctx <-> one HIP stream <-> one host thread.  Distinct ctxs are
 *     independent (the reference's class-level singletons are not).
 *   - Return value 0 = ok, <0 = error (PA_E_*); pa_last_error() gives the text.
 *     Non-convergence is not an error (report.converge = 0), as in the reference.
 *   - All work is enqueued on the ctx stream; only the solver entry points and
 *     pa_report_read synchronise the stream.
 */
#ifndef PYAPES_HIP_H
#define PYAPES_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pa_ctx pa_ctx;

enum { PA_F32 = 0, PA_F64 = 1 };

/* boundary types: pyapes/variables/bcs.py:197-280, BC_FACTORY :460 */
enum { PA_BC_NONE = 0, PA_BC_DIRICHLET = 1, PA_BC_NEUMANN = 2, PA_BC_SYMMETRY = 3, PA_BC_PERIODIC = 4 };

/* operator kinds: pyapes/solver/fdc.py Laplacian :369, Grad :461, Div :612 */
enum {
  PA_OP_LAPLACIAN = 0,
  PA_OP_GRAD = 1,          /* solver use is 1-D only (ops.py:145-147) */
  PA_OP_DIV_CENTRAL = 2,   /* limiter "none"  (fdc.py:708-743) */
  PA_OP_DIV_UPWIND_COMPAT = 3, /* limiter "upwind", literal reference output (fdc.py:746-772) */
  PA_OP_DIV_UPWIND = 4     /* first-order upwind as the reference's test states it (tests/test_fdm.py:239) */
};

enum {
  PA_OK = 0,
  PA_E_ARG = -1,       /* bad argument / unsupported configuration */
  PA_E_HIP = -2,       /* HIP runtime error */
  PA_E_STATE = -3,     /* call order (grid / equation not set) */
  PA_E_NONFINITE = -4  /* stop-test value became NaN/Inf: RuntimeError in linalg.py:334-336 */
};

/* pyapes/solver/linalg.py:22-30 ReportType, plus timing/diagnostics (non-breaking) */
typedef struct {
  int64_t itr;       /* iterations executed */
  double tol;        /* last stop-test value */
  int32_t converge;  /* itr < max_it */
  int32_t status;    /* PA_OK or PA_E_NONFINITE */
  double rr;         /* last sum r.r (diagnostic) */
  double gpu_ms;     /* stream time of the solve loop (HIP events) */
} pa_report;

/* one term of  sum_k sign_k * param_k * Op_k(x)   (pyapes/solver/ops.py:122-154, fdm.py OPStype) */
typedef struct {
  int32_t kind;          /* PA_OP_* */
  int32_t has_coeff;     /* laplacian/grad: param is not None (fdm.py:166-169) */
  double sign;           /* +1 / -1 (fdm.py:95-105) */
  double coeff;          /* scalar param */
  const void* coeff_field; /* tensor param (device, field-shaped) or NULL */
  double u;              /* div: scalar advection speed (fdc.py:778-779) */
  const void* u_field;   /* div: advection tensor (device, field-shaped) or NULL */
} pa_term;

/* ---- context ---------------------------------------------------------- */
/* replaces: device/dtype gate Mesh(..., device, dtype) pyapes/mesh/_mesh.py:30-44, backend.py:7-94 */
int pa_ctx_create(int device, void* hip_stream, pa_ctx** out);
int pa_ctx_destroy(pa_ctx* ctx);
/* Kernel-path switches of one ctx (tests and A/B measurements; defaults from PYAPES_HIP_FASTPATH / _SF /
 * _FOLD / _RESIDENT at pa_ctx_create): "fastpath" 0 = generic kernels only, "sf" 0 = k_cg3d instead of k_sf
 * for the Div-carrying single-field operations, "fold" 0 = scalar steps as single-block kernels, "resident"
 * 0 = launch-per-phase solver loops on small meshes too.  The first three do not change results
 * (bit-identical paths; tests/test_gpu_properties.py, test_gpu_fold.py); "resident" changes the grouping of
 * the global sums only (tests/test_gpu_resident.py).  Round 3: "pitch" 0 = odd row lengths on the one-cell-per-lane
 * kernels (else pitched ctx-owned buffers), "cg2d_mincells" = 2-D marching kernel from this many cells on (< 0:
 * never), "bcl" 0 = Euler march with a BC fill per step, "place" 0 = no online placement search in large CG solves
 * (which allocations r / d live in beside the caller's x: csrc/pa_place.hip; results do not depend on it,
 * tests/test_gpu_place.py) with "place_minbytes" (arrays of at least this size; default 128 MiB), "place_blocks"
 * (candidate allocations per role, default 3) and "place_budget" (what the trials may cost, per cent of the time
 * solved so far, default 3), "resident_coop" 0 = plain launch of the resident kernel (profiling). */
int pa_ctx_set_option(pa_ctx* ctx, const char* name, int value);
/* Further names (tests and A/B runs; none has an environment variable of its own, PYAPES_HIP_OPTIONS="name=value,..." sets
 * any option of this list for every ctx of the process): "bc_path" (bit 0: never the closed-form BC fill, bit 1: never
 * the per-axis pair kernels, bit 2: closed form at any size; all paths bit-identical), "bicg_pfold" 0 = BiCGSTAB with the
 * full p / v phase in every iteration, "bicg_srv" 0 = BiCGSTAB's tiled s / t phase stores s as well (1: t alone, the x / r
 * update re-forms s = r - alpha v', the same bits), "jac_alt" 0 = every Jacobi sweep of a 3-D mesh marches forwards (1: consecutive
 * sweeps in opposite directions; the iterates are the same bits, the stop-test sums are added in another order), "rhs_full" 1 = pa_rhs_adjust over the whole mesh, "res_cells" / "res_nt" /
 * "res_nt_cells" / "res_spin" / "res_rzlean" (box plan, threads per workgroup, spin bound, rz stencil of the resident
 * solver), "comm" / "slab_fold" (read back by the slab driver: library-side RCCL loop, folded sequence), "comm_overlap"
 * (-1 auto), "comm_timeout" (s), "roctx" (process-wide).  pa_ctx_get_option reads a value back. */
int pa_ctx_get_option(const pa_ctx* ctx, const char* name, int* value);
/* Accounts of the online placement search of this ctx, out[10]: state (-1 off, 0 not begun / arrays too small,
 * 1 searching, 2 pass over), trials, trials kept, allocations made, microseconds the trials have cost (copies,
 * memsets, trial iterations that ran slower), microseconds of iterations timed, best iteration pair (us), blocks
 * held at the moment, the first clean iteration pair of the context (as allocated, us), x pointers with a finished pass. */
int pa_place_stats(pa_ctx* ctx, double* out);
/* Small meshes: pa_cg / pa_jacobi / pa_bicgstab run the whole solve in ONE cooperative launch with the fields
 * resident in LDS (pa_resident.hip) when the bound mesh / BCs / equation allow it.  Returns the number of
 * workgroups (= boxes the mesh is cut into; boxes[3] = boxes per internal axis) such a solve would use
 * (solver: 0 CG, 1 Jacobi, 2 BiCGSTAB), 0 when the launch-per-phase loops would run.  pa_resident_used: the
 * same for the LAST solve of this ctx. */
int pa_resident_plan(pa_ctx* ctx, int solver, int* boxes);
int pa_resident_used(const pa_ctx* ctx);
/* The stream every later call enqueues on (the binding passes torch's CURRENT stream before each call, so
 * that work the caller queued on it -- tensor allocation, input preparation -- is ordered before the
 * kernels).  Refused while a stepwise solve is live on another stream. */
int pa_ctx_set_stream(pa_ctx* ctx, void* hip_stream);
const char* pa_last_error(const pa_ctx* ctx); /* ctx may be NULL: error of the failed create */
const char* pa_version(void);

/* ---- grid -------------------------------------------------------------
 * replaces: Mesh node counts / spacing (pyapes/mesh/_mesh.py:67-93, 258-298).
 * n[]: LOCAL node counts; dx[]: spacing exactly as stored in the mesh dtype.
 * Slab decomposition (new): this rank owns global planes
 * [i_off, i_off + n[0]) of n0_global along axis 0.  Single GPU: i_off = 0,
 * n0_global = n[0]. */
int pa_grid_set(pa_ctx* ctx, int ndim, const int64_t* n, const double* dx, int dtype,
                int64_t i_off, int64_t n0_global);

/* ---- coordinate system ------------------------------------------------
 * replaces: the axisymmetric (Cylinder, "rz") rows of default_A_ops (pyapes/solver/tools.py:64-107),
 * Laplacian.build_A_coeffs / adjust_rhs (fdc.py:395-417, 440-453).  2-D grids only (axis 0 = r,
 * axis 1 = z; _mesh.py:48-49).  r_nodes: device pointer to the n[0] node radii (Mesh.x[0]) in the grid
 * dtype; only read during the call.  The library evaluates, once, the r-dependent coefficient rows
 * with the reference's literal expressions (dr = dx[0]; nn = nan_to_num(., 0, 0, 0)):
 *   (1 +- nn(dr / (2 r))) / dr^2     Laplacian Ap / Am along r
 *   (2/3 + nn(2/3 dr / r)) / dr^2    neumann / symmetry row along r (Ap = -Ac or Am = -Ac)
 *   2/3 - nn(1/3 dr / r)             factor of the Laplacian rhs adjustment along r
 *   nn(2 dr / r)                     Ac of Div along r (the u phi / r term)
 * PA_COORD_XYZ (r_nodes ignored) restores the Cartesian operators; pa_grid_set resets to it. */
enum { PA_COORD_XYZ = 0, PA_COORD_RZ = 1 };
int pa_coord_set(pa_ctx* ctx, int coord_sys, const void* r_nodes);

/* ---- boundary conditions ---------------------------------------------
 * replaces: BC objects + BC.apply (pyapes/variables/bcs.py:70-95, 200-280) and the
 * list order of Field.bcs (fields.py:378-416).  order_pos = position in the
 * application list.  value/face_vals: Dirichlet value g, or Neumann gradient V,
 * scalar or per-node array in boolean-mask gather order (= C order of the face
 * plane); face_vals is a device pointer the caller keeps alive, or NULL.
 * dxf = grid[face] - grid[prev] (the literal coordinate difference used by
 * Neumann.apply, bcs.py:228-231); ignored for the other types. */
int pa_bc_set(pa_ctx* ctx, int face, int order_pos, int type, double value,
              const void* face_vals, double dxf);
int pa_bc_clear(pa_ctx* ctx);
/* linalg._apply_bc_otf (linalg.py:282-299): all faces, in order, in place */
int pa_apply_bc(pa_ctx* ctx, void* x);

/* ---- equation ---------------------------------------------------------
 * replaces: the ops dict built by fdm.laplacian/grad/div(...) and +,-,neg
 * (pyapes/solver/fdm.py:75-105, 124-312) consumed by ops._Aop (ops.py:122-154) */
int pa_eq_set(pa_ctx* ctx, int nterms, const pa_term* terms);
/* y = sum_k sign_k param_k Op_k(x).  interior_only=0: every node, wrap-around
 * neighbours like torch.roll (fdc.py:171-200); 1: interior set S only
 * (mesh/tools.py:7-20), y = 0 elsewhere. */
int pa_aop(pa_ctx* ctx, const void* x, void* y, int interior_only);
/* rhs += sum_k rhs_adj_k  (Solver.set_eq, ops.py:63-77; fdc.py:426-458, 505-540, 667-694) */
int pa_rhs_adjust(pa_ctx* ctx, void* rhs);

/* ---- explicit operators (Discretizer.__call__/apply, fdc.py:67-168) ---- */
int pa_laplacian(pa_ctx* ctx, const void* x, void* y, int edge);
/* y is (ndim, n...) */
int pa_grad(pa_ctx* ctx, const void* x, void* y, int edge);
/* kind = PA_OP_DIV_*; u_field NULL -> scalar u */
int pa_div(pa_ctx* ctx, int kind, double u, const void* u_field, const void* x, void* y);
/* edge=True post-pass of Div (fdc.py:290-361): overwrites the two end nodes of y with the one-sided
 * 2nd-order formula times the advection value.  1-D only, like the reference for scalar fields. */
int pa_div_edge(pa_ctx* ctx, double u, const void* u_field, const void* x, void* y);

/* ---- general Div, DiffFlux and the Fokker-Planck operators ------------------
 * pa_div_general replaces Div.build_A_coeffs + Discretizer.apply + _treat_edge in full generality
 * (fdc.py:93-102, 290-361, 623-664, 708-772): y = sum over mesh axes a of the stencil along a applied
 * to x[a] (scalar field: the same pointer for every a; vector field: component a) with the advection
 * u_int[a] (field or NULL = the scalar u; the reference uses component 0 / Jac[n2d[0]] on every axis
 * for a scalar target and component a for a vector target -- the binding resolves that); with
 * edge != 0 the two end planes of axis a of that contribution are replaced by the one-sided formula
 * times u_edge[a] (or u), plus the literal rz terms on the r axis.  Single GPU.
 * pa_diff_flux replaces DiffFlux.__call__ (fdc.py:818-856): out[i] = sum_j w_i D[i*ndim+j] J[j],
 * w = r on the r row of an rz mesh, out shaped (ndim, *n).
 * pa_rfp_friction / pa_rfp_diffusion replace Friction / Diffusion.__call__ (solver/rfp.py:19-82,
 * 85-218), rz grids only.  pa_limiter: minmod (0) / mc_limiter (1), rfp.py:262-286, n elements. */
typedef struct {
  const void* x[3];
  const void* u_int[3];
  const void* u_edge[3];
  double u;
  int kind;  /* PA_OP_DIV_CENTRAL | PA_OP_DIV_UPWIND_COMPAT | PA_OP_DIV_UPWIND */
  int edge;
} pa_div_spec;
int pa_div_general(pa_ctx* ctx, const pa_div_spec* spec, void* y);
int pa_diff_flux(pa_ctx* ctx, const void* const* D, const void* const* J, void* out);
int pa_rfp_friction(pa_ctx* ctx, const void* Hr, const void* Hz, const void* pdf, void* out);
int pa_rfp_diffusion(pa_ctx* ctx, const void* Drr, const void* Drz, const void* Dzz, const void* pdf, void* out);
int pa_limiter(pa_ctx* ctx, int which, const void* a, const void* b, void* out, int64_t n);

/* ---- solvers (linalg.solve -> cg | bicgstab, linalg.py:33-279) --------- */
/* replaces: var.save_old() at the top of every solver iteration (linalg.py:110, 210; fields.py:129-131):
 * x_old = device field the solvers below keep equal to the iterate BEFORE the last executed iteration
 * (what Field.VARo holds after linalg.solve); NULL (default) = not kept, no extra pass. */
int pa_solver_keep_old(pa_ctx* ctx, void* x_old);
int pa_cg(pa_ctx* ctx, void* x, const void* rhs, double tol, int64_t max_it, pa_report* out);
int pa_bicgstab(pa_ctx* ctx, void* x, const void* rhs, double tol, int64_t max_it, pa_report* out);
/* new: weighted Jacobi with the CG's BC fill / interior set / stop test */
int pa_jacobi(pa_ctx* ctx, void* x, const void* rhs, double tol, int64_t max_it, double omega,
              pa_report* out);
/* new: phi_out = B(phi + dt (nu lap(phi) - div(u phi))) on the interior set.
 * In slab mode (pa_slab_set) the step reads the ghost planes x_ghost_lo / hi (NULL = physical end) and leaves the BC
 * fill B to the driver (pyapes_amd/slab.py SlabEuler: exchange first / last plane of phi -> step -> on a periodic axis 0
 * exchange the far planes of phi_out -> pa_apply_bc). */
int pa_euler_step(pa_ctx* ctx, const void* phi_in, void* phi_out, int div_kind, double u,
                  const void* u_field, double nu, double dt);

/* nsteps Euler steps enqueued back to back, ping-ponging phi <-> tmp (final state in phi if nsteps
 * is even, else in tmp); no host synchronisation */
int pa_euler_march(pa_ctx* ctx, void* phi, void* tmp, int div_kind, double u, const void* u_field,
                   double nu, double dt, int64_t nsteps);

/* ---- vector steps for a host-stepped solver loop (pyapes_amd/solver/host_stepped.py: BC callables that read the iterate
 * make the reference's loop come back to Python for every face of every fill, bcs.py:200-253; the loop's AXPYs and dot
 * products between those fills are these).  Fields of the grid's shape and dtype.
 *   pa_vec_axpy  out = y + a x, the product rounded before the sum (linalg.py:120, 127, 138; "-": pass -a); out may be y or x
 *   pa_vec_dot   *result = sum a.b (diff 0) or sum (a - b)^2 (diff 1: the stop test's norm, linalg.py:321-338), products in
 *                the grid dtype, summed in double; synchronises */
int pa_vec_axpy(pa_ctx* ctx, void* out, const void* y, double a, const void* x);
int pa_vec_dot(pa_ctx* ctx, const void* a, const void* b, int diff, double* result);
/* x <- 0 off the interior set of the BC list bound last (r = b - A x lives on it, linalg.py:99-101) */
int pa_vec_mask_interior(pa_ctx* ctx, void* x);

/* ---- stepwise CG (slab-decomposed multi-GPU driver and bench.py) -------
 * pa_cg above is the single-GPU loop.  The stepwise form lets a host driver put
 * the halo exchange and the scalar all-reduces (RCCL through torch.distributed)
 * between the phases of one iteration:
 *   begin -> [exchange r planes] -> [all-reduce sums] -> per iteration:
 *     phase_a   d' = r + beta d, local sum d'.(A d')        -> [all-reduce sums]
 *     phase_b   alpha; x += alpha d'; r -= alpha A d'   -> [exchange r planes, periodic x planes]
 *     bc        BC fill of x, boundary part of the stop test, local sums -> [all-reduce sums]
 *     finish_iter   beta, stop test, iteration count (device side)
 * With no slab set (P = 1) the BC fill, reductions and scalar logic run inside the phases
 * and bc / finish_iter are no-ops.  On a slab pa_cg_begin does NOT fill the BCs: the driver
 * calls pa_apply_bc first and then exchanges the ghost planes of x.  Nothing here
 * synchronises the stream. */
#define PA_NSUM 8
enum { PA_SUM_DAD = 0, PA_SUM_RR = 1, PA_SUM_DX2 = 2 };

/* Exchange buffers of a slab rank; all device pointers owned by the caller
 * (torch tensors), n1*n2 elements per plane.  NULL recv pointer = physical
 * (non-periodic) end of the domain on that side. */
typedef struct {
  void* sums;              /* PA_NSUM doubles: local partial sums out, all-reduced sums in */
  void* r_send_lo;         /* out: first owned plane of r (after begin / phase_b) */
  void* r_send_hi;         /* out: last owned plane of r */
  const void* r_recv_lo;   /* in: ghost plane of r below the slab */
  const void* r_recv_hi;   /* in: ghost plane of r above the slab */
  const void* x_ghost_lo;  /* in: ghost planes of x for the initial residual / pa_aop */
  const void* x_ghost_hi;
  const void* bc_far_lo0;  /* periodic axis-0 BC fill on the lower end rank: x[N-1], x[N-2] */
  const void* bc_far_lo1;
  const void* bc_far_hi0;  /* ... on the upper end rank: x[1] of the lower end rank (the new x[0] is
                              recomputed there bit for bit as x[1] - x[N-1] + x[N-2]) */
  /* optional packing of the periodic x planes next to the residual planes, so that one message per
   * neighbour carries everything: after phase_b the library copies x[1] to x_pack_lo1 (lower end
   * rank) and x[n0-1], x[n0-2] to x_pack_hi0 / x_pack_hi1 (upper end rank); NULL = not wanted */
  void* x_pack_lo1;
  void* x_pack_hi0;
  void* x_pack_hi1;
} pa_slab;
int pa_slab_set(pa_ctx* ctx, const pa_slab* slab); /* NULL: back to single GPU */

int pa_cg_begin(pa_ctx* ctx, void* x, const void* rhs, double tol, int64_t max_it);
int pa_cg_phase_a(pa_ctx* ctx);
int pa_cg_phase_b(pa_ctx* ctx);
int pa_cg_bc(pa_ctx* ctx);
int pa_cg_finish_iter(pa_ctx* ctx);
/* P = 1 only: enqueue n whole iterations back to back.  Inside the batch the scalar step between two
 * kernels (alpha; beta, stop test, iteration count) runs in the prologue of the kernel that follows
 * instead of in a single-block kernel of its own; the batch's last step is flushed before the call
 * returns, so pa_report_read / pa_cg_end see the same state as after n x { phase_a, phase_b }. */
int pa_cg_iterate(pa_ctx* ctx, int64_t n);
int pa_cg_end(pa_ctx* ctx, pa_report* out);       /* synchronises */
int pa_cg_abort(pa_ctx* ctx);                     /* drop the live stepwise solve (no read-back) */
/* ---- stepwise BiCGSTAB on a slab (linalg.py:162-279 split at its reductions and exchanges; SURVEY 8e + 8f-1) ----
 * BASELINE config 3 is periodic: CG never meets the reference's stop test there, BiCGSTAB does -- so it has
 * to exist on slabs.  A host driver (pyapes_amd/slab.py SlabBiCGSTAB) puts the communication between the calls:
 *   [pa_apply_bc, exchange ghost planes of x]
 *   begin     r0 = r = b - A x on S, local r0.r0 -> sums[1]; r planes out  -> [exchange r planes] [all-reduce sums[1]]
 *   start     rho' = r0.r0, first beta (device side)
 *   per iteration:
 *     pv      p' = r + beta (p - omega v) (ghost planes by the owner's recurrence: p is never sent), v' = A p' on S,
 *             local r0.v' -> sums[0]; v' planes out                        -> [all-reduce sums[0]] [exchange v' planes]
 *     st      alpha; s = r - alpha v', t = A s on S; local |s|^2, t.s, t.t, r0.t -> sums[1..4] -> [all-reduce sums[1..4]]
 *     x       stop test 1, omega, rho'; x += alpha p' + omega s, r = s - omega t; r planes (+ periodic x planes) out
 *                                                                           -> [exchange r planes (+ x planes)]
 *     bc      BC fill of x; local |r|^2 -> sums[5]                          -> [all-reduce sums[5]]
 *     finish  stop test 2, beta, rho <- rho' (device side)
 *   end       report (synchronises)
 * pa_slab_set_v: the planes of v' (n1 * n2 elements each; NULL receive pointer = physical end); the other planes are
 * those of pa_slab_set.  Nothing here synchronises the stream; iterations enqueued after the stop are no-ops. */
int pa_slab_set_v(pa_ctx* ctx, void* v_send_lo, void* v_send_hi, const void* v_recv_lo, const void* v_recv_hi);
int pa_bicg_begin(pa_ctx* ctx, void* x, const void* rhs, double tol, int64_t max_it);
int pa_bicg_start(pa_ctx* ctx);
int pa_bicg_pv(pa_ctx* ctx);
int pa_bicg_st(pa_ctx* ctx);
int pa_bicg_x(pa_ctx* ctx);
int pa_bicg_bc(pa_ctx* ctx);
int pa_bicg_finish(pa_ctx* ctx);
int pa_bicg_end(pa_ctx* ctx, pa_report* out);
/* ---- stepwise Jacobi on a slab (the sweep of pa_jacobi, SURVEY a15, split at its exchanges and its reduction) ----
 * The iterate ping-pongs between the caller's x and a field of the context, so the planes the neighbours need leave
 * through the buffers of pa_slab_set: the periodic far planes of the new iterate through x_pack_*, its first / last
 * owned plane (BCs filled) through r_send_lo / hi -- the driver (pyapes_amd/slab.py SlabJacobi) delivers the latter
 * INTO the neighbours' x_ghost_lo / hi, which every sweep reads.
 *   [pa_apply_bc, exchange ghost planes of x]
 *   begin   records the start's boundary shell (x_old of the first stop test)
 *   per sweep:
 *     sweep   x' = x + omega (b - A x) / diag(A) on S; periodic far planes of x' out     -> [exchange far planes]
 *     bc      BC fill of x'; local |x' - x|^2 -> sums[2]; first / last plane of x' out    -> [all-reduce sums[2]]
 *                                                                                          [planes -> neighbours' x_ghost]
 *     finish  stop test, sweep count (device side); the buffers change roles
 *   end     report (synchronises); the final iterate is copied into x if it lives in the context's field
 * Laplacian terms only, like pa_jacobi (a zero diagonal otherwise). */
int pa_jacobi_begin(pa_ctx* ctx, void* x, const void* rhs, double tol, int64_t max_it, double omega);
int pa_jacobi_sweep(pa_ctx* ctx);
int pa_jacobi_bc(pa_ctx* ctx);
int pa_jacobi_finish(pa_ctx* ctx);
int pa_jacobi_end(pa_ctx* ctx, pa_report* out);
/* While a stepwise solve is live (pa_cg_begin ... pa_cg_end / pa_cg_abort) pa_bc_clear, pa_bc_set and
 * pa_eq_set return PA_E_STATE: every phase re-reads that state. */
int pa_report_read(pa_ctx* ctx, pa_report* out);  /* synchronises */
/* The solver's scalar state as the last executed iteration left it (the reference computes these inline and
 * drops them, linalg.py:118-141 / 205-262; here they live on the device): out[PA_NSCALAR] =
 * { alpha, beta, r.r, r.r of the iteration before, d.Ad, tol, rho, omega, rho_next, r0.v, t.s, t.t, r0.t, itr, 0, 0 }
 * (CG uses the first six, BiCGSTAB alpha + the rho ... r0.t group).  Parity tests compare them, iteration by
 * iteration, with the reference's. */
#define PA_NSCALAR 16
int pa_scalars_read(pa_ctx* ctx, double* out);

/* ---- RCCL inside the library (slab mode, SURVEY 8e: "pa_comm_init(ctx, rank, nranks, rccl_unique_id)") --
 * The stepwise calls above let a host driver run the exchange; these entry points put the whole
 * iteration -- kernels, the two scalar all-reduces (ncclAllReduce on the PA_NSUM buffer) and the
 * one packed plane exchange per neighbour (grouped ncclSend / ncclRecv) -- on the ctx stream, so
 * that n iterations are enqueued by ONE call with no host work in between.  librccl is resolved at
 * run time (dlopen of the copy already in the process, e.g. PyTorch's); without it these return
 * PA_E_STATE and the stepwise path remains.
 *   pa_comm_unique_id   rank 0: 128-byte ncclUniqueId to hand to the other ranks (any transport)
 *   pa_comm_init        collective: ncclCommInitRank on the ctx device
 *   pa_comm_selftest    collective: a sum all-reduce and a ring exchange with known answers, the
 *                       stream polled against timeout_s; on timeout the communicator is aborted
 *                       and an error returned (callers then fall back to the stepwise path)
 *   pa_comm_plan        neighbours (-1 = none) and the packed send / receive buffers of one
 *                       iteration, counts in elements of the grid dtype; lower neighbour first
 *   pa_cg_iterate_comm  n x { phase_a, all-reduce sums[0], phase_b, exchange, bc,
 *                             all-reduce sums[1..2], finish_iter }   (stepwise sequence)
 *
 * Folded iterations.  Per iteration the stepwise sequence costs six single-block / per-plane launches
 * beside the two phases (ghost-direction recurrence, two reductions of per-workgroup partial rows, alpha,
 * plane packing, beta + stop test).  When every rank runs the tiled kernels, the ranks agree on row
 * counts (pa_cg_fold_plan on each rank after pa_cg_begin, MAX over the ranks by the host driver,
 * pa_cg_fold_set) and pa_cg_iterate_comm then all-reduces the partial ROWS (a few KB, the same
 * latency-bound collective) and lets the prologues of the kernels that follow sum them:
 *   phase_a (closes the previous iteration in its prologue) -> all-reduce d.Ad rows -> mid kernel
 *   (alpha; ghost planes of the new direction; the residual / x planes the neighbours need, computed
 *   ahead of phase B) -> packed exchange on a second communicator + stream, beside: phase_b -> BC fill
 *   -> all-reduce (r.r, |dx|^2, shell) rows.
 * Same arithmetic per node; the sums differ from the stepwise sequence only in the order ranks and rows
 * are added (identical for one rank).  rows[3] = { phase A, phase B, BC shell }; all zero = not applicable
 * on this rank (then every rank must stay stepwise: pa_cg_fold_set with zeros or not at all). */
int pa_cg_fold_plan(pa_ctx* ctx, int64_t* rows3);
int pa_cg_fold_set(pa_ctx* ctx, const int64_t* rows3);

typedef struct {
  int nb_lo, nb_hi;
  const void* send_lo; const void* send_hi;
  void* recv_lo; void* recv_hi;
  int64_t n_send_lo, n_send_hi, n_recv_lo, n_recv_hi;
} pa_exchange;
int pa_comm_available(void);                  /* 1: librccl resolved in this process (no communicator needed) */
int pa_comm_unique_id(void* id128);
int pa_comm_count(pa_ctx* ctx, int* nranks); /* ncclCommCount of the library's communicator */
int pa_comm_init(pa_ctx* ctx, int rank, int nranks, const void* id128);
int pa_comm_selftest(pa_ctx* ctx, double timeout_s);
int pa_comm_plan(pa_ctx* ctx, const pa_exchange* plan);
int pa_cg_iterate_comm(pa_ctx* ctx, int64_t n);
/* the same for the stepwise BiCGSTAB (pa_bicg_begin ... pa_bicg_end): n iterations = 5 step calls + 3 all-reduces + the
 * exchanges of the v' planes (pa_slab_set_v) and of the packed r / x planes (pa_comm_plan) each, one C call */
int pa_bicg_iterate_comm(pa_ctx* ctx, int64_t n);
/* and for the stepwise Jacobi (pa_jacobi_begin ... pa_jacobi_end): n sweeps = 3 step calls + 1 all-reduce + the plane
 * exchange(s) each, one C call */
int pa_jacobi_iterate_comm(pa_ctx* ctx, int64_t n);
int pa_comm_destroy(pa_ctx* ctx);
/* Give up on the library's communicators while work that uses them may still be queued (a collective some rank
 * never joined): ncclCommAbort on both, then the ctx streams are drained.  The stepwise calls remain usable. */
int pa_comm_abort(pa_ctx* ctx);
/* Wait until everything enqueued on the ctx stream(s) has run, at most timeout_s seconds: PA_OK, or PA_E_STATE
 * when work is still queued (watchdog of multi-rank drivers: bench.py bounds its first slab iterations with it) */
int pa_stream_wait(pa_ctx* ctx, double timeout_s);
/* Which implementation is behind pa_comm_*: "rccl" (librccl, resolved at run time), "none", or "custom: ..." after
 * pa_comm_use_impl. */
const char* pa_comm_impl(void);
/* Take the twelve nccl* entry points from the shared library at so_path instead of librccl.  An explicit call, to be made
 * before the first communicator of the process; nothing in the environment selects an implementation.  Exists for
 * tests: tests/lib/libpa_hostring.so runs the ranks as processes that share one GPU (RCCL refuses that), with host
 * shared memory as the wire. */
int pa_comm_use_impl(const char* so_path);
/* 1 when the plane exchange of pa_cg_iterate_comm runs on the second communicator + stream, 0 when on the ctx stream */
int pa_comm_overlap(const pa_ctx* ctx);

/* ---- measurement (bench.py roofline leg) ---------------------------------
 * on: bracket each launch of the two dominant CG kernels (phase A stencil+dot,
 * phase B update+stencil+dots) with HIP events on the ctx stream and accumulate
 * their durations (the host waits per launch: measurement loops only). */
int pa_profile_set(pa_ctx* ctx, int on);
int pa_profile_read(pa_ctx* ctx, double* ms_phase_a, int64_t* n_phase_a, double* ms_phase_b,
                    int64_t* n_phase_b);

#ifdef __cplusplus
}
#endif
#endif /* PYAPES_HIP_H */
