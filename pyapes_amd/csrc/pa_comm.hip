// pa_comm.hip -- RCCL inside the library: the slab-decomposed CG iteration enqueued on
// the ctx stream with its two scalar all-reduces and its one packed plane exchange per neighbour,
// no host work between the phases.  librccl is resolved at run time: the copy already mapped into
// the process (PyTorch's) if there is one, else the system's; the library itself does not link it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>

#include "../../include/pyapes_hip.h"
#include "pa_device.h"
#include "pa_comm_table.h"
#include "pa_host.h"

namespace {

Rccl g_rccl;

template <typename F>
bool sym(void* h, const char* name, F& f) {
  f = reinterpret_cast<F>(dlsym(h, name));
  return f != nullptr;
}

bool resolve(void* h, Rccl& R) {
  return sym(h, "ncclGetUniqueId", R.GetUniqueId) && sym(h, "ncclCommInitRank", R.CommInitRank) &&
         sym(h, "ncclCommDestroy", R.CommDestroy) && sym(h, "ncclCommAbort", R.CommAbort) && sym(h, "ncclCommCount", R.CommCount) &&
         sym(h, "ncclAllReduce", R.AllReduce) && sym(h, "ncclBroadcast", R.Broadcast) && sym(h, "ncclSend", R.Send) && sym(h, "ncclRecv", R.Recv) &&
         sym(h, "ncclGroupStart", R.GroupStart) && sym(h, "ncclGroupEnd", R.GroupEnd) &&
         sym(h, "ncclGetErrorString", R.GetErrorString);
}

Rccl* rccl() {
  Rccl& R = g_rccl;
  if (R.tried) return R.h ? &R : nullptr;
  R.tried = 1;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);  // the copy the process already uses
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) return nullptr;
  if (!resolve(h, R)) return nullptr;
  R.h = h;
  return &R;
}

int nccl_fail(pa_ctx* c, Rccl* R, ncclResult_t e, const char* what) {
  pa_set_err(c, "RCCL: %s failed: %s", what, R->GetErrorString ? R->GetErrorString(e) : "?");
  return PA_E_HIP;
}

#define PA_NCCL(c, R, call)                                     \
  do {                                                          \
    ncclResult_t e_ = (call);                                   \
    if (e_ != ncclSuccess) return nccl_fail((c), (R), e_, #call); \
  } while (0)

// wait for the ctx stream with a deadline; false = still busy
bool stream_done_within(hipStream_t s, double timeout_s) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    hipError_t q = hipStreamQuery(s);
    if (q == hipSuccess) return true;
    if (q != hipErrorNotReady) { (void)hipGetLastError(); return false; }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
    std::this_thread::sleep_for(std::chrono::milliseconds(2));
  }
}

int exchange(pa_ctx* c, Rccl* R, ncclComm_t comm, hipStream_t st) {
  const pa_exchange& P = c->plan;
  const ncclDataType_t dt = c->dtype == PA_F64 ? ncclDouble : ncclFloat;
  if (P.nb_lo < 0 && P.nb_hi < 0) return PA_OK;
  // order: sends [down, up], receives [from above, from below] -- with P = 2 on a periodic ring both
  // neighbours are the same peer and same-peer operations pair up in program order
  PA_NCCL(c, R, R->GroupStart());
  if (P.nb_lo >= 0 && P.n_send_lo > 0) PA_NCCL(c, R, R->Send(P.send_lo, (size_t)P.n_send_lo, dt, P.nb_lo, comm, st));
  if (P.nb_hi >= 0 && P.n_send_hi > 0) PA_NCCL(c, R, R->Send(P.send_hi, (size_t)P.n_send_hi, dt, P.nb_hi, comm, st));
  if (P.nb_hi >= 0 && P.n_recv_hi > 0) PA_NCCL(c, R, R->Recv(P.recv_hi, (size_t)P.n_recv_hi, dt, P.nb_hi, comm, st));
  if (P.nb_lo >= 0 && P.n_recv_lo > 0) PA_NCCL(c, R, R->Recv(P.recv_lo, (size_t)P.n_recv_lo, dt, P.nb_lo, comm, st));
  PA_NCCL(c, R, R->GroupEnd());
  return PA_OK;
}

// the boundary planes of v' (stepwise BiCGSTAB, pa_slab_set_v): same neighbours, same order as exchange()
int exchange_v(pa_ctx* c, Rccl* R, ncclComm_t comm, hipStream_t st) {
  const pa_exchange& P = c->plan;
  const ncclDataType_t dt = c->dtype == PA_F64 ? ncclDouble : ncclFloat;
  const size_t n = (size_t)c->G.s0;
  if (P.nb_lo < 0 && P.nb_hi < 0) return PA_OK;
  PA_NCCL(c, R, R->GroupStart());
  if (P.nb_lo >= 0 && c->v_send_lo) PA_NCCL(c, R, R->Send(c->v_send_lo, n, dt, P.nb_lo, comm, st));
  if (P.nb_hi >= 0 && c->v_send_hi) PA_NCCL(c, R, R->Send(c->v_send_hi, n, dt, P.nb_hi, comm, st));
  if (P.nb_hi >= 0 && c->v_recv_hi) PA_NCCL(c, R, R->Recv((void*)c->v_recv_hi, n, dt, P.nb_hi, comm, st));
  if (P.nb_lo >= 0 && c->v_recv_lo) PA_NCCL(c, R, R->Recv((void*)c->v_recv_lo, n, dt, P.nb_lo, comm, st));
  PA_NCCL(c, R, R->GroupEnd());
  return PA_OK;
}

// stepwise Jacobi: (which = 0) the periodic far planes of the new iterate, i.e. the packed buffers WITHOUT their first
// plane, between the end ranks of the ring; (which = 1) the first / last owned plane of the new iterate into the
// neighbours' ghost planes of x (pa_slab_set's x_ghost_lo / hi).  Same neighbours and same order as exchange().
int exchange_jacobi(pa_ctx* c, Rccl* R, ncclComm_t comm, hipStream_t st, int which) {
  const pa_exchange& P = c->plan;
  const ncclDataType_t dt = c->dtype == PA_F64 ? ncclDouble : ncclFloat;
  const int64_t n = c->G.s0;
  if (P.nb_lo < 0 && P.nb_hi < 0) return PA_OK;
  const size_t off = (size_t)n * (size_t)c->esize;
  const void* s_lo = which ? c->r_send_lo : (const void*)((const char*)P.send_lo + off);
  const void* s_hi = which ? c->r_send_hi : (const void*)((const char*)P.send_hi + off);
  void* r_lo = which ? (void*)c->x_glo : (void*)((char*)P.recv_lo + off);
  void* r_hi = which ? (void*)c->x_ghi : (void*)((char*)P.recv_hi + off);
  const int64_t ns_lo = which ? (c->r_send_lo ? n : 0) : P.n_send_lo - n, ns_hi = which ? (c->r_send_hi ? n : 0) : P.n_send_hi - n;
  const int64_t nr_lo = which ? (c->x_glo ? n : 0) : P.n_recv_lo - n, nr_hi = which ? (c->x_ghi ? n : 0) : P.n_recv_hi - n;
  if (ns_lo <= 0 && ns_hi <= 0 && nr_lo <= 0 && nr_hi <= 0) return PA_OK;
  PA_NCCL(c, R, R->GroupStart());
  if (P.nb_lo >= 0 && ns_lo > 0) PA_NCCL(c, R, R->Send(s_lo, (size_t)ns_lo, dt, P.nb_lo, comm, st));
  if (P.nb_hi >= 0 && ns_hi > 0) PA_NCCL(c, R, R->Send(s_hi, (size_t)ns_hi, dt, P.nb_hi, comm, st));
  if (P.nb_hi >= 0 && nr_hi > 0) PA_NCCL(c, R, R->Recv(r_hi, (size_t)nr_hi, dt, P.nb_hi, comm, st));
  if (P.nb_lo >= 0 && nr_lo > 0) PA_NCCL(c, R, R->Recv(r_lo, (size_t)nr_lo, dt, P.nb_lo, comm, st));
  PA_NCCL(c, R, R->GroupEnd());
  return PA_OK;
}

}  // namespace

extern "C" {

int pa_comm_unique_id(void* id128) {
  Rccl* R = rccl();
  if (!R || !id128) return PA_E_STATE;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  if (R->GetUniqueId(&id) != ncclSuccess) return PA_E_HIP;
  memcpy(id128, &id, sizeof(id));
  return PA_OK;
}

int pa_comm_available(void) { return rccl() ? 1 : 0; }

int pa_comm_count(pa_ctx* c, int* nranks) {
  if (!c || !nranks) return PA_E_ARG;
  Rccl* R = rccl();
  if (!R || !c->comm) { pa_set_err(c, "pa_comm_count without pa_comm_init"); return PA_E_STATE; }
  PA_NCCL(c, R, R->CommCount((ncclComm_t)c->comm, nranks));
  return PA_OK;
}

int pa_comm_init(pa_ctx* c, int rank, int nranks, const void* id128) {
  if (!c || !id128) return PA_E_ARG;
  Rccl* R = rccl();
  if (!R) { pa_set_err(c, "pa_comm_init: librccl not available in this process"); return PA_E_STATE; }
  if (c->comm) { pa_set_err(c, "pa_comm_init: communicator already set"); return PA_E_STATE; }
  if (nranks < 1 || rank < 0 || rank >= nranks) { pa_set_err(c, "pa_comm_init: bad rank %d / %d", rank, nranks); return PA_E_ARG; }
  PA_HIP(c, hipSetDevice(c->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t comm = nullptr;
  PA_NCCL(c, R, R->CommInitRank(&comm, nranks, id, rank));
  c->comm = comm;
  c->comm_rank = rank;
  c->comm_n = nranks;
  c->plan_set = 0;
  // Second communicator on its own stream for the plane exchange (PYAPES_HIP_COMM_OVERLAP=0: none).  Its id
  // travels through the first one; every step that can fail on one rank alone is followed by an agreement
  // (MIN all-reduce of an ok flag on the first communicator) before any rank depends on the result.
  // Default: only when there is a peer.  A 1-rank communicator "exchanges" with itself through HBM, and a
  // copy kernel beside phase B is starved by it (measured: 14.5 us alone, 114 us beside phase B on a
  // 64 x 512^2 slab): there is no link latency to hide.  PYAPES_HIP_COMM_OVERLAP=1 / 0 forces either.
  int want = c->comm_overlap >= 0 ? c->comm_overlap : (nranks > 1);   // option "comm_overlap" / PYAPES_HIP_COMM_OVERLAP
  // Host side of every bounded wait below is PINNED: a device-to-host copy into pageable memory blocks the host
  // until the stream reaches it, i.e. inside the very collective whose completion is in doubt -- the deadline
  // would never be looked at -- and a copy that is still queued when this frame is left would land in a dead
  // stack frame.  hb lives until the stream has been waited for (or drained after an abort).
  struct Box { ncclUniqueId id; int ok; int all; int pad[2]; };
  Box* hb = nullptr;
  Box* dev = nullptr;
  const double tmo = (double)c->comm_timeout;   // option "comm_timeout" / PYAPES_HIP_COMM_TIMEOUT
  // leave with the first communicator aborted: a rank that cannot go on must not keep a communicator the others
  // are about to give up on (they do so after `tmo`; the caller's agreement then puts every rank on the stepwise path)
  auto give_up = [&](const char* why) {
    (void)R->CommAbort(comm);
    c->comm = nullptr;
    (void)hipStreamSynchronize(c->stream);   // nothing queued may still write to hb / dev
    if (dev) (void)hipFree(dev);
    if (hb) (void)hipHostFree(hb);
    (void)hipGetLastError();
    pa_set_err(c, "pa_comm_init: %s (limit %.0f s); communicator aborted", why, tmo);
    return PA_E_STATE;
  };
  if (hipHostMalloc((void**)&hb, sizeof(Box), hipHostMallocDefault) != hipSuccess || hipMalloc((void**)&dev, sizeof(Box)) != hipSuccess)
    return give_up("no memory for the set-up of the second communicator");
  memset(hb, 0, sizeof(Box));
  if (rank == 0) {
    hb->ok = (want && R->GetUniqueId(&hb->id) == ncclSuccess) ? 1 : 0;
    if (hipMemcpyAsync(dev, hb, sizeof(Box), hipMemcpyHostToDevice, c->stream) != hipSuccess) return give_up("hipMemcpyAsync failed");
  }
  if (R->Broadcast(dev, dev, sizeof(Box), ncclChar, 0, comm, c->stream) != ncclSuccess ||
      hipMemcpyAsync(hb, dev, sizeof(Box), hipMemcpyDeviceToHost, c->stream) != hipSuccess)
    return give_up("broadcast of the second communicator's id could not be enqueued");
  if (!stream_done_within(c->stream, tmo)) return give_up("no completion of the id broadcast");
  if (hb->ok) {
    ncclComm_t comm2 = nullptr;
    int prio_lo = 0, prio_hi = 0;   // highest priority: its own hardware queue, and its few workgroups first
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    int ok = want && R->CommInitRank(&comm2, nranks, hb->id, rank) == ncclSuccess &&
             hipStreamCreateWithPriority(&c->xstream, hipStreamNonBlocking, prio_hi) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_x, hipEventDisableTiming) == hipSuccess;
    auto drop2 = [&]() {
      if (c->xstream) { (void)hipStreamDestroy(c->xstream); c->xstream = nullptr; }
      if (c->ev_b) { (void)hipEventDestroy(c->ev_b); c->ev_b = nullptr; }
      if (c->ev_x) { (void)hipEventDestroy(c->ev_x); c->ev_x = nullptr; }
    };
    hb->ok = ok;
    hb->all = 0;
    bool queued = hipMemcpyAsync(&dev->ok, &hb->ok, sizeof(int), hipMemcpyHostToDevice, c->stream) == hipSuccess &&
                  R->AllReduce(&dev->ok, &dev->ok, 1, ncclInt, ncclMin, comm, c->stream) == ncclSuccess &&
                  hipMemcpyAsync(&hb->all, &dev->ok, sizeof(int), hipMemcpyDeviceToHost, c->stream) == hipSuccess;
    if (!queued || !stream_done_within(c->stream, tmo)) {
      if (comm2) (void)R->CommAbort(comm2);
      drop2();
      return give_up(queued ? "no completion of the agreement on the second communicator" : "agreement could not be enqueued");
    }
    if (hb->all) {
      c->comm2 = comm2;
    } else {
      if (comm2) (void)R->CommDestroy(comm2);
      drop2();
    }
  }
  (void)hipGetLastError();
  (void)hipFree(dev);
  (void)hipHostFree(hb);
  return PA_OK;
}

int pa_comm_abort(pa_ctx* c) {
  if (!c) return PA_E_ARG;
  Rccl* R = rccl();
  if (R) {   // kernels of a collective that will never complete return once their communicator is aborted
    if (c->comm2) (void)R->CommAbort((ncclComm_t)c->comm2);
    if (c->comm) (void)R->CommAbort((ncclComm_t)c->comm);
  }
  c->comm2 = nullptr;
  c->comm = nullptr;
  if (c->xstream) (void)hipStreamSynchronize(c->xstream);
  (void)hipStreamSynchronize(c->stream);
  if (c->xstream) { (void)hipStreamDestroy(c->xstream); c->xstream = nullptr; }
  if (c->ev_b) { (void)hipEventDestroy(c->ev_b); c->ev_b = nullptr; }
  if (c->ev_x) { (void)hipEventDestroy(c->ev_x); c->ev_x = nullptr; }
  (void)hipGetLastError();
  c->plan_set = 0;
  c->slab_fold_live = 0;
  return PA_OK;
}

int pa_stream_wait(pa_ctx* c, double timeout_s) {
  if (!c) return PA_E_ARG;
  if (stream_done_within(c->stream, timeout_s) && (!c->xstream || stream_done_within(c->xstream, timeout_s))) return PA_OK;
  pa_set_err(c, "pa_stream_wait: work still queued after %.1f s", timeout_s);
  return PA_E_STATE;
}

int pa_comm_overlap(const pa_ctx* c) { return c && c->comm && c->comm2 && c->xstream ? 1 : 0; }

const char* pa_comm_impl(void) {
  Rccl* R = rccl();
  return R ? R->impl : "none";
}

// Another provider of the twelve nccl* entry points than librccl: an explicit call, made before the first communicator
// of the process exists, with the path of a shared library that exports them (tests: tests/lib/libpa_hostring.so, ranks as
// processes sharing one GPU).  No environment variable selects anything here.
int pa_comm_use_impl(const char* so_path) {
  Rccl& R = g_rccl;
  if (!so_path || !so_path[0]) return PA_E_ARG;
  if (R.tried && R.h) return PA_E_STATE;   // librccl (or another library) is in use already
  void* h = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "[pyapes_hip] pa_comm_use_impl: %s\n", dlerror()); return PA_E_ARG; }
  Rccl T;
  if (!resolve(h, T)) { dlclose(h); return PA_E_ARG; }
  R = T;
  R.tried = 1;
  R.h = h;
  const char* (*name)() = nullptr;
  snprintf(R.impl_buf, sizeof(R.impl_buf), "custom: %s", sym(h, "pa_comm_impl_name", name) ? name() : so_path);
  R.impl = R.impl_buf;
  fprintf(stderr, "[pyapes_hip] collectives of this process come from %s, not from librccl\n", so_path);
  return PA_OK;
}

int pa_comm_destroy(pa_ctx* c) {
  if (!c) return PA_E_ARG;
  Rccl* R = rccl();
  if (c->comm && R) {
    (void)hipStreamSynchronize(c->stream);
    if (c->xstream) (void)hipStreamSynchronize(c->xstream);
    if (c->comm2) (void)R->CommDestroy((ncclComm_t)c->comm2);
    (void)R->CommDestroy((ncclComm_t)c->comm);
  }
  if (c->xstream) { (void)hipStreamDestroy(c->xstream); c->xstream = nullptr; }
  if (c->ev_b) { (void)hipEventDestroy(c->ev_b); c->ev_b = nullptr; }
  if (c->ev_x) { (void)hipEventDestroy(c->ev_x); c->ev_x = nullptr; }
  c->comm2 = nullptr;
  c->comm = nullptr;
  c->plan_set = 0;
  return PA_OK;
}

int pa_comm_selftest(pa_ctx* c, double timeout_s) {
  if (!c || !c->comm) { if (c) pa_set_err(c, "pa_comm_selftest without pa_comm_init"); return PA_E_STATE; }
  Rccl* R = rccl();
  ncclComm_t comm = (ncclComm_t)c->comm;
  PA_HIP(c, hipSetDevice(c->device));
  double* dev = nullptr;
  PA_HIP(c, hipMalloc(&dev, 4 * sizeof(double)));
  const int n = c->comm_n, rk = c->comm_rank;
  double h[4] = {1.0, (double)(rk + 1), (double)rk, -1.0};
  if (hipError_t he = hipMemcpyAsync(dev, h, sizeof(h), hipMemcpyHostToDevice, c->stream); he != hipSuccess) {
    (void)hipFree(dev);
    return pa_hip_fail(c, he, "pa_comm_selftest: hipMemcpyAsync");
  }
  ncclResult_t e = R->AllReduce(dev, dev, 2, ncclDouble, ncclSum, comm, c->stream);
  if (e == ncclSuccess) e = R->GroupStart();
  if (e == ncclSuccess) e = R->Send(dev + 2, 1, ncclDouble, (rk + 1) % n, comm, c->stream);
  if (e == ncclSuccess) e = R->Recv(dev + 3, 1, ncclDouble, (rk + n - 1) % n, comm, c->stream);
  if (e == ncclSuccess) e = R->GroupEnd();
  if (e == ncclSuccess && c->comm2) {   // the exchange communicator: same ring on its own stream, joined below
    ncclComm_t c2 = (ncclComm_t)c->comm2;
    (void)hipEventRecord(c->ev_b, c->stream);
    (void)hipStreamWaitEvent(c->xstream, c->ev_b, 0);
    e = R->GroupStart();
    if (e == ncclSuccess) e = R->Send(dev + 2, 1, ncclDouble, (rk + 1) % n, c2, c->xstream);
    if (e == ncclSuccess) e = R->Recv(dev + 3, 1, ncclDouble, (rk + n - 1) % n, c2, c->xstream);
    if (e == ncclSuccess) e = R->GroupEnd();
    (void)hipEventRecord(c->ev_x, c->xstream);
    (void)hipStreamWaitEvent(c->stream, c->ev_x, 0);
  }
  if (e != ncclSuccess) { (void)hipFree(dev); return nccl_fail(c, R, e, "selftest enqueue"); }
  if (!stream_done_within(c->stream, timeout_s)) {
    if (c->comm2) { (void)R->CommAbort((ncclComm_t)c->comm2); c->comm2 = nullptr; }
    (void)R->CommAbort(comm);
    c->comm = nullptr;
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(dev);
    pa_set_err(c, "pa_comm_selftest: no completion within %.1f s, communicator aborted", timeout_s);
    return PA_E_STATE;
  }
  const hipError_t back = hipMemcpy(h, dev, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(dev);
  if (back != hipSuccess) return pa_hip_fail(c, back, "pa_comm_selftest: hipMemcpy");
  const double want1 = 0.5 * n * (n + 1.0), want3 = (double)((rk + n - 1) % n);
  if (h[0] != (double)n || h[1] != want1 || h[3] != want3) {
    pa_set_err(c, "pa_comm_selftest: wrong answers (%g %g %g, expected %d %g %g)", h[0], h[1], h[3], n, want1, want3);
    return PA_E_STATE;
  }
  return PA_OK;
}

int pa_comm_plan(pa_ctx* c, const pa_exchange* plan) {
  if (!c || !plan) return PA_E_ARG;
  if (!c->comm) { pa_set_err(c, "pa_comm_plan without pa_comm_init"); return PA_E_STATE; }
  const int n = c->comm_n;
  if (plan->nb_lo >= n || plan->nb_hi >= n) { pa_set_err(c, "pa_comm_plan: neighbour outside the communicator"); return PA_E_ARG; }
  if ((plan->nb_lo >= 0 && ((plan->n_send_lo > 0 && !plan->send_lo) || (plan->n_recv_lo > 0 && !plan->recv_lo))) ||
      (plan->nb_hi >= 0 && ((plan->n_send_hi > 0 && !plan->send_hi) || (plan->n_recv_hi > 0 && !plan->recv_hi)))) {
    pa_set_err(c, "pa_comm_plan: null buffer with a non-zero count");
    return PA_E_ARG;
  }
  c->plan = *plan;
  c->plan_set = 1;
  return PA_OK;
}

int pa_cg_iterate_comm(pa_ctx* c, int64_t n) {
  if (!c || !c->solver_live) { if (c) pa_set_err(c, "pa_cg_iterate_comm without pa_cg_begin"); return PA_E_STATE; }
  if (!c->slab || !c->ext_sums) { pa_set_err(c, "pa_cg_iterate_comm needs slab mode (pa_slab_set)"); return PA_E_STATE; }
  if (!c->comm || !c->plan_set) { pa_set_err(c, "pa_cg_iterate_comm needs pa_comm_init + pa_comm_plan"); return PA_E_STATE; }
  PA_HIP(c, hipSetDevice(c->device));
  Rccl* R = rccl();
  ncclComm_t comm = (ncclComm_t)c->comm;
  double* sums = c->ext_sums;
  int rc;
  if (!c->slab_fold || c->profile) {
    // stepwise sequence (also the per-kernel timing loop of pa_profile_set): sums are all-reduced, every
    // scalar step is its own single-block kernel
    for (int64_t q = 0; q < n; ++q) {
      if ((rc = pa_place_tick(c))) return rc;   // every rank searches for itself (the slowest sets the pace of all)
      if ((rc = pa_cg_phase_a(c))) return rc;
      PA_NCCL(c, R, R->AllReduce(sums + PA_SUM_DAD, sums + PA_SUM_DAD, 1, ncclDouble, ncclSum, comm, c->stream));
      if ((rc = pa_cg_phase_b(c))) return rc;
      if ((rc = exchange(c, R, comm, c->stream))) return rc;
      if ((rc = pa_cg_bc(c))) return rc;
      PA_NCCL(c, R, R->AllReduce(sums + PA_SUM_RR, sums + PA_SUM_RR, 2, ncclDouble, ncclSum, comm, c->stream));
      if ((rc = pa_cg_finish_iter(c))) return rc;
    }
    return pa_place_batch_end(c);
  }
  // folded sequence (include/pyapes_hip.h "Folded iterations"): 2 tiled kernels + the mid kernel + the BC
  // fill + 2 row all-reduces on the ctx stream, the packed exchange beside them on its own communicator +
  // stream.  Rows: [ A | S | B ]; the first all-reduce carries A, the second S + B.
  const size_t mA = (size_t)c->fold_rows[0], mSB = (size_t)c->fold_rows[2] + 2 * (size_t)c->fold_rows[1];
  const bool side = c->comm2 && c->xstream;
  // does THIS rank's BC fill read planes that arrive with the exchange (end ranks of a periodic ring)?
  const bool bc_needs_x = c->bc_far_lo0 || c->bc_far_lo1 || c->bc_far_hi0;
  const bool any_nb = c->plan.nb_lo >= 0 || c->plan.nb_hi >= 0;
#define PA_RT(call) if ((call) != hipSuccess) { rc = PA_E_HIP; break; }
#define PA_NC(call) if ((call) != ncclSuccess) { rc = PA_E_HIP; break; }
  c->slab_fold_live = 1;
  rc = PA_OK;
  for (int64_t q = 0; q < n && !rc; ++q) {
    if ((rc = pa_place_tick(c))) break;
    if ((rc = pa_cg_phase_a(c))) break;
    PA_NC(R->AllReduce(c->rows_send, c->rows_recv, mA, ncclDouble, ncclSum, comm, c->stream));
    if ((rc = pa_cg_slab_mid(c))) break;
    bool x_pending = false;
    if (any_nb) {
      if (side) {
        PA_RT(hipEventRecord(c->ev_b, c->stream));
        PA_RT(hipStreamWaitEvent(c->xstream, c->ev_b, 0));
        if ((rc = exchange(c, R, (ncclComm_t)c->comm2, c->xstream))) break;
        PA_RT(hipEventRecord(c->ev_x, c->xstream));
        x_pending = true;
      } else if ((rc = exchange(c, R, comm, c->stream))) {
        break;
      }
    }
    if ((rc = pa_cg_phase_b(c))) break;
    if (x_pending && bc_needs_x) {
      PA_RT(hipStreamWaitEvent(c->stream, c->ev_x, 0));
      x_pending = false;
    }
    if ((rc = pa_cg_bc(c))) break;
    PA_NC(R->AllReduce(c->rows_send + mA, c->rows_recv + mA, mSB, ncclDouble, ncclSum, comm, c->stream));
    if (x_pending) PA_RT(hipStreamWaitEvent(c->stream, c->ev_x, 0));
  }
#undef PA_RT
#undef PA_NC
  // the batch's last scalar step (beta, stop test, iteration count) by the single-block kernel the prologue
  // of a next phase A would have replaced: pa_report_read / pa_cg_end see the state of n whole iterations
  if (!rc) rc = pa_place_batch_end(c);
  if (!rc) rc = pa_cg_slab_flush(c);
  c->slab_fold_live = 0;
  if (rc == PA_E_HIP && !c->err[0]) pa_set_err(c, "pa_cg_iterate_comm: HIP / RCCL call failed in the folded sequence");
  return rc;
}

// n BiCGSTAB iterations on a slab with everything on the ctx stream: the five step calls of the stepwise form
// (include/pyapes_hip.h), the three small all-reduces and the two plane exchanges between them.  Iterations enqueued
// after the device-side stop are no-ops (the collectives still run: every rank enqueues the same sequence).
int pa_bicg_iterate_comm(pa_ctx* c, int64_t n) {
  if (!c || c->solver_live != 2) { if (c) pa_set_err(c, "pa_bicg_iterate_comm without pa_bicg_begin"); return PA_E_STATE; }
  if (!c->slab || !c->ext_sums) { pa_set_err(c, "pa_bicg_iterate_comm needs slab mode (pa_slab_set)"); return PA_E_STATE; }
  if (!c->comm || !c->plan_set) { pa_set_err(c, "pa_bicg_iterate_comm needs pa_comm_init + pa_comm_plan"); return PA_E_STATE; }
  PA_HIP(c, hipSetDevice(c->device));
  Rccl* R = rccl();
  ncclComm_t comm = (ncclComm_t)c->comm;
  double* sums = c->ext_sums;
  int rc;
  for (int64_t q = 0; q < n; ++q) {
    if ((rc = pa_bicg_pv(c))) return rc;
    PA_NCCL(c, R, R->AllReduce(sums + 0, sums + 0, 1, ncclDouble, ncclSum, comm, c->stream));
    if ((rc = exchange_v(c, R, comm, c->stream))) return rc;
    if ((rc = pa_bicg_st(c))) return rc;
    PA_NCCL(c, R, R->AllReduce(sums + 1, sums + 1, 4, ncclDouble, ncclSum, comm, c->stream));
    if ((rc = pa_bicg_x(c))) return rc;
    if ((rc = exchange(c, R, comm, c->stream))) return rc;
    if ((rc = pa_bicg_bc(c))) return rc;
    PA_NCCL(c, R, R->AllReduce(sums + 5, sums + 5, 1, ncclDouble, ncclSum, comm, c->stream));
    if ((rc = pa_bicg_finish(c))) return rc;
  }
  return PA_OK;
}

// n Jacobi sweeps on a slab, everything on the ctx stream: sweep -> [far planes of a periodic ring] -> BC fill + local
// |dx|^2 -> all-reduce -> first / last plane into the neighbours' ghost planes -> stop test.  Sweeps enqueued after the
// device-side stop are no-ops (the collectives still run).
int pa_jacobi_iterate_comm(pa_ctx* c, int64_t n) {
  if (!c || c->solver_live != 3) { if (c) pa_set_err(c, "pa_jacobi_iterate_comm without pa_jacobi_begin"); return PA_E_STATE; }
  if (!c->slab || !c->ext_sums) { pa_set_err(c, "pa_jacobi_iterate_comm needs slab mode (pa_slab_set)"); return PA_E_STATE; }
  if (!c->comm || !c->plan_set) { pa_set_err(c, "pa_jacobi_iterate_comm needs pa_comm_init + pa_comm_plan"); return PA_E_STATE; }
  PA_HIP(c, hipSetDevice(c->device));
  Rccl* R = rccl();
  ncclComm_t comm = (ncclComm_t)c->comm;
  double* sums = c->ext_sums;
  int rc;
  for (int64_t q = 0; q < n; ++q) {
    if ((rc = pa_jacobi_sweep(c))) return rc;
    if ((rc = exchange_jacobi(c, R, comm, c->stream, 0))) return rc;
    if ((rc = pa_jacobi_bc(c))) return rc;
    PA_NCCL(c, R, R->AllReduce(sums + PA_SUM_DX2, sums + PA_SUM_DX2, 1, ncclDouble, ncclSum, comm, c->stream));
    if ((rc = exchange_jacobi(c, R, comm, c->stream, 1))) return rc;
    if ((rc = pa_jacobi_finish(c))) return rc;
  }
  return PA_OK;
}

}  // extern "C"
