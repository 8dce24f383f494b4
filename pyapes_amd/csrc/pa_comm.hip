// pa_comm.hip -- RCCL inside the library: the slab-decomposed CG iteration enqueued on
// the ctx stream with its two scalar all-reduces and its one packed plane exchange per neighbour,
// no host work between the phases.  librccl is resolved at run time: the copy already mapped into
// the process (PyTorch's) if there is one, else the system's; the library itself does not link it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstring>
#include <thread>

#include "../../include/pyapes_hip.h"
#include "pa_device.h"
#include "pa_host.h"

namespace {

struct Rccl {
  void* h = nullptr;
  int tried = 0;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;

template <typename F>
bool sym(void* h, const char* name, F& f) {
  f = reinterpret_cast<F>(dlsym(h, name));
  return f != nullptr;
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
  bool ok = sym(h, "ncclGetUniqueId", R.GetUniqueId) && sym(h, "ncclCommInitRank", R.CommInitRank) &&
            sym(h, "ncclCommDestroy", R.CommDestroy) && sym(h, "ncclCommAbort", R.CommAbort) && sym(h, "ncclCommCount", R.CommCount) &&
            sym(h, "ncclAllReduce", R.AllReduce) && sym(h, "ncclSend", R.Send) && sym(h, "ncclRecv", R.Recv) &&
            sym(h, "ncclGroupStart", R.GroupStart) && sym(h, "ncclGroupEnd", R.GroupEnd) &&
            sym(h, "ncclGetErrorString", R.GetErrorString);
  if (!ok) return nullptr;
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

int exchange(pa_ctx* c, Rccl* R) {
  const pa_exchange& P = c->plan;
  const ncclDataType_t dt = c->dtype == PA_F64 ? ncclDouble : ncclFloat;
  ncclComm_t comm = (ncclComm_t)c->comm;
  if (P.nb_lo < 0 && P.nb_hi < 0) return PA_OK;
  // order: sends [down, up], receives [from above, from below] -- with P = 2 on a periodic ring both
  // neighbours are the same peer and same-peer operations pair up in program order
  PA_NCCL(c, R, R->GroupStart());
  if (P.nb_lo >= 0 && P.n_send_lo > 0) PA_NCCL(c, R, R->Send(P.send_lo, (size_t)P.n_send_lo, dt, P.nb_lo, comm, c->stream));
  if (P.nb_hi >= 0 && P.n_send_hi > 0) PA_NCCL(c, R, R->Send(P.send_hi, (size_t)P.n_send_hi, dt, P.nb_hi, comm, c->stream));
  if (P.nb_hi >= 0 && P.n_recv_hi > 0) PA_NCCL(c, R, R->Recv(P.recv_hi, (size_t)P.n_recv_hi, dt, P.nb_hi, comm, c->stream));
  if (P.nb_lo >= 0 && P.n_recv_lo > 0) PA_NCCL(c, R, R->Recv(P.recv_lo, (size_t)P.n_recv_lo, dt, P.nb_lo, comm, c->stream));
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
  return PA_OK;
}

int pa_comm_destroy(pa_ctx* c) {
  if (!c) return PA_E_ARG;
  Rccl* R = rccl();
  if (c->comm && R) {
    (void)hipStreamSynchronize(c->stream);
    (void)R->CommDestroy((ncclComm_t)c->comm);
  }
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
  if (e != ncclSuccess) { (void)hipFree(dev); return nccl_fail(c, R, e, "selftest enqueue"); }
  if (!stream_done_within(c->stream, timeout_s)) {
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
  Rccl* R = rccl();
  ncclComm_t comm = (ncclComm_t)c->comm;
  double* sums = c->ext_sums;
  int rc;
  for (int64_t q = 0; q < n; ++q) {
    if ((rc = pa_cg_phase_a(c))) return rc;
    PA_NCCL(c, R, R->AllReduce(sums + PA_SUM_DAD, sums + PA_SUM_DAD, 1, ncclDouble, ncclSum, comm, c->stream));
    if ((rc = pa_cg_phase_b(c))) return rc;
    if ((rc = exchange(c, R))) return rc;
    if ((rc = pa_cg_bc(c))) return rc;
    PA_NCCL(c, R, R->AllReduce(sums + PA_SUM_RR, sums + PA_SUM_RR, 2, ncclDouble, ncclSum, comm, c->stream));
    if ((rc = pa_cg_finish_iter(c))) return rc;
  }
  return PA_OK;
}

}  // extern "C"
