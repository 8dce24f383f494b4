// pa_hostring.hip -- TEST-ONLY stand-in for the RCCL entry points libpyapes_hip uses (csrc/pa_comm_table.h), built into
// tests/lib/libpa_hostring.so.  It is NOT part of the product library: a test hands its path to pa_comm_use_impl(), which
// resolves the twelve nccl* symbols from it exactly as it resolves them from librccl.
//
// Why it exists: RCCL refuses two ranks on one device, and the builder's box has one GPU, so the N > 1 form of
// the library-side slab loop (pa_cg_iterate_comm: grouped send / recv between distinct peers, the same-peer P = 2
// ring, out-of-place row all-reduces with uneven slabs, the cross-stream event pair of the second communicator)
// could not execute anywhere before the driver's multi-GPU run.  With this library selected (pa_comm_use_impl: an explicit
// call, never implicit) the SAME C code of pa_comm.hip runs with 2 / 4 rank PROCESSES that
// share the one GPU: real kernels, real k_slab_mid, real streams and events; only the wire is replaced.
//
// Semantics kept from RCCL: every call ENQUEUES on the caller's stream and returns; data is read / written in
// stream order; same-peer send / recv pair up in program order; a group's sends and receives progress together;
// an all-reduce returns the same bits on every rank (ranks added in rank order).  The wire: a POSIX shared-memory
// segment per communicator (named by the unique id).  Per operation k of a communicator:
//   stream:  k_hr_post   copies the send buffer into a pinned staging slot, then releases posted = k + 1
//   helper thread (one per communicator, never calls HIP): waits for posted, moves the slot through the shared
//            segment (all-reduce: every rank's slot -> sum in rank order; send: a 2-deep FIFO per ordered pair;
//            recv: the FIFO's head), fills the pinned receive slot, releases ready = k + 1
//   stream:  k_hr_fetch  spins (bounded) on ready, copies the slot into the receive buffer, releases fetched
// Every wait -- host, helper, device -- is bounded; an expired wait marks the communicator failed, still releases
// whatever waits on it (garbage data, loud message on stderr) and nothing hangs.
// Fault injection for the fall-back tests: PYAPES_HIP_HOSTRING_FAIL="<rank>:<what>:<index>", what =
//   init     the <index>-th ncclCommInitRank of that rank's process fails (after the ranks have met)
//   corrupt  the <index>-th fp64 all-reduce on that rank's FIRST communicator returns wrong numbers
//   hang     ... never completes (until the communicator is aborted or the helper's wait expires)
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

namespace {

constexpr int HR_MAXR = 8;
constexpr size_t HR_AR_CAP = 256u << 10;    // bytes of one all-reduce / broadcast contribution
constexpr size_t HR_P2P_CAP = 8u << 20;     // bytes of one message (a plane of BASELINE config 3 / 5 is 2 MiB, the periodic ring sends three at once)
constexpr int HR_P2P_DEPTH = 2;
constexpr int HR_NS = 8;                    // staging slots (operations in flight) per communicator
constexpr size_t HR_STAGE_CAP = HR_P2P_CAP;
enum { F_POSTED = 0, F_READY = 8, F_FETCHED = 16, F_STATUS = 24, F_WORDS = 32 };   // one 64-byte line each
enum { OP_AR = 0, OP_BCAST, OP_SEND, OP_RECV };

struct alignas(64) HrFifo {
  std::atomic<uint64_t> head;   // messages posted by the sender
  std::atomic<uint64_t> tail;   // messages consumed by the receiver
  uint64_t len[HR_P2P_DEPTH];
};

struct HrShm {
  alignas(64) std::atomic<uint32_t> attached;
  alignas(64) std::atomic<uint64_t> ar_posted[HR_MAXR][8];   // [q][0]: collectives rank q has contributed to
  HrFifo fifo[HR_MAXR][HR_MAXR];                             // [src][dst]
  // then: ar_data[HR_MAXR][2][HR_AR_CAP], p2p_data[HR_MAXR][HR_MAXR][HR_P2P_DEPTH][HR_P2P_CAP] (sparse: only the
  // pages a test touches exist)
};
constexpr size_t HR_HDR = (sizeof(HrShm) + 4095) & ~size_t(4095);
constexpr size_t HR_AR_BYTES = (size_t)HR_MAXR * 2 * HR_AR_CAP;
constexpr size_t HR_BYTES = HR_HDR + HR_AR_BYTES + (size_t)HR_MAXR * HR_MAXR * HR_P2P_DEPTH * HR_P2P_CAP;

struct HrOp {
  int kind = 0, peer = 0, root = 0;
  size_t bytes = 0, count = 0;
  ncclDataType_t dt = ncclChar;
  ncclRedOp_t op = ncclSum;
  bool has_in = false;
};

struct HrComm {
  int rank = 0, n = 0, index = 0;   // index: which communicator of this process (0 = the first)
  HrShm* shm = nullptr;
  char *ar_data = nullptr, *p2p_data = nullptr;
  char *stage_out = nullptr, *stage_in = nullptr;   // pinned
  uint64_t* flags = nullptr;                        // pinned, F_WORDS
  uint64_t enq = 0;                                 // host thread of the communicator's owner
  HrOp ring[HR_NS];
  std::mutex mu;
  std::condition_variable cv;
  uint64_t enq_pub = 0;
  bool stop = false;
  std::atomic<uint64_t> helper_done{0};
  std::atomic<int> abort{0}, failed{0};
  std::thread helper;
  std::vector<hipStream_t> streams;
  double timeout_s = 60.0;
  uint64_t coll_seq = 0, ar_f64 = 0;   // helper thread only
  int fail_what = 0;                   // 0 none, 2 corrupt, 3 hang
  long fail_index = -1;
};

int g_init_calls = 0;
int g_comms_made = 0;

inline uint64_t ld_acq(const uint64_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline void st_rel(uint64_t* p, uint64_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }

size_t dt_size(ncclDataType_t dt) {
  switch (dt) {
    case ncclChar: case ncclUint8: return 1;
    case ncclInt: case ncclUint32: case ncclFloat: return 4;
    case ncclInt64: case ncclUint64: case ncclDouble: return 8;
    default: return 0;
  }
}

template <typename P>
bool spin_until(HrComm* c, P pred, const char* what) {
  const auto t0 = std::chrono::steady_clock::now();
  for (int it = 0;; ++it) {
    if (pred()) return true;
    if (c->abort.load(std::memory_order_relaxed)) return false;
    if ((it & 63) == 63) {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) {
        fprintf(stderr, "[pyapes hostring] rank %d/%d comm %d: no progress within %.0f s while %s\n", c->rank, c->n, c->index,
                c->timeout_s, what);
        return false;
      }
      std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
  }
}

template <typename T>
void reduce_ranks(T* out, const HrComm* c, size_t count, ncclRedOp_t op, int parity) {
  for (int q = 0; q < c->n; ++q) {   // ranks in rank order: the same bits on every rank
    const T* src = reinterpret_cast<const T*>(c->ar_data + ((size_t)q * 2 + parity) * HR_AR_CAP);
    if (q == 0) {
      memcpy(out, src, count * sizeof(T));
    } else if (op == ncclMin) {
      for (size_t i = 0; i < count; ++i) out[i] = src[i] < out[i] ? src[i] : out[i];
    } else {
      for (size_t i = 0; i < count; ++i) out[i] = out[i] + src[i];
    }
  }
}

void helper_main(HrComm* c) {
  for (uint64_t k = 0;; ++k) {
    {
      std::unique_lock<std::mutex> lk(c->mu);
      c->cv.wait(lk, [&] { return c->enq_pub > k || c->stop; });
      if (c->enq_pub <= k) break;
    }
    const HrOp op = c->ring[k % HR_NS];
    char* out = c->stage_out + (k % HR_NS) * HR_STAGE_CAP;
    char* in = c->stage_in + (k % HR_NS) * HR_STAGE_CAP;
    bool ok = !c->failed.load();
    const bool has_out = op.kind != OP_RECV;
    if (ok && has_out) ok = spin_until(c, [&] { return ld_acq(c->flags + F_POSTED) >= k + 1; }, "waiting for the stream to stage a send buffer");
    if (op.kind == OP_AR || op.kind == OP_BCAST) {
      const uint64_t s = c->coll_seq++;
      const int par = (int)(s & 1);
      bool hang = false, corrupt = false;
      if (op.kind == OP_AR && op.dt == ncclDouble && c->index == 0) {
        const long j = (long)c->ar_f64++;
        hang = c->fail_what == 3 && j == c->fail_index;
        corrupt = c->fail_what == 2 && j == c->fail_index;
      }
      if (ok && hang) {
        fprintf(stderr, "[pyapes hostring] rank %d: injected hang in fp64 all-reduce %ld\n", c->rank, c->fail_index);
        ok = spin_until(c, [] { return false; }, "hanging on purpose (PYAPES_HIP_HOSTRING_FAIL)");
      }
      if (ok) {
        if (op.kind == OP_AR || c->rank == op.root) memcpy(c->ar_data + ((size_t)c->rank * 2 + par) * HR_AR_CAP, out, op.bytes);
        c->shm->ar_posted[c->rank][0].store(s + 1, std::memory_order_release);
        for (int q = 0; q < c->n && ok; ++q)
          ok = spin_until(c, [&] { return c->shm->ar_posted[q][0].load(std::memory_order_acquire) >= s + 1; }, "waiting for a rank to join a collective");
      }
      if (ok) {
        if (op.kind == OP_BCAST) {
          memcpy(in, c->ar_data + ((size_t)op.root * 2 + par) * HR_AR_CAP, op.bytes);
        } else if (op.dt == ncclDouble) {
          reduce_ranks(reinterpret_cast<double*>(in), c, op.count, op.op, par);
          if (corrupt) reinterpret_cast<double*>(in)[0] += 1.0;
        } else if (op.dt == ncclFloat) {
          reduce_ranks(reinterpret_cast<float*>(in), c, op.count, op.op, par);
        } else if (op.dt == ncclInt) {
          reduce_ranks(reinterpret_cast<int*>(in), c, op.count, op.op, par);
        } else {
          ok = false;
        }
      }
    } else if (op.kind == OP_SEND) {
      HrFifo& f = c->shm->fifo[c->rank][op.peer];
      char* base = c->p2p_data + (((size_t)c->rank * HR_MAXR + op.peer) * HR_P2P_DEPTH) * HR_P2P_CAP;
      if (ok) ok = spin_until(c, [&] { return f.head.load(std::memory_order_relaxed) - f.tail.load(std::memory_order_acquire) < HR_P2P_DEPTH; }, "waiting for room in a peer's FIFO");
      if (ok) {
        const uint64_t h = f.head.load(std::memory_order_relaxed);
        memcpy(base + (h % HR_P2P_DEPTH) * HR_P2P_CAP, out, op.bytes);
        f.len[h % HR_P2P_DEPTH] = op.bytes;
        f.head.store(h + 1, std::memory_order_release);
      }
    } else {
      HrFifo& f = c->shm->fifo[op.peer][c->rank];
      char* base = c->p2p_data + (((size_t)op.peer * HR_MAXR + c->rank) * HR_P2P_DEPTH) * HR_P2P_CAP;
      if (ok) ok = spin_until(c, [&] { return f.head.load(std::memory_order_acquire) > f.tail.load(std::memory_order_relaxed); }, "waiting for a peer's message");
      if (ok) {
        const uint64_t t = f.tail.load(std::memory_order_relaxed);
        if (f.len[t % HR_P2P_DEPTH] != op.bytes) {
          fprintf(stderr, "[pyapes hostring] rank %d: message of %llu bytes from rank %d where %zu were expected\n", c->rank,
                  (unsigned long long)f.len[t % HR_P2P_DEPTH], op.peer, op.bytes);
          ok = false;
        } else {
          memcpy(in, base + (t % HR_P2P_DEPTH) * HR_P2P_CAP, op.bytes);
        }
        f.tail.store(t + 1, std::memory_order_release);
      }
    }
    if (!ok) {
      c->failed.store(1);
      st_rel(c->flags + F_STATUS, k + 1);
    }
    st_rel(c->flags + F_READY, k + 1);   // always: a fetch kernel must never wait for an operation that failed
    c->helper_done.store(k + 1, std::memory_order_release);
  }
}

// ---- device side -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void hr_copy(char* dst, const char* src, size_t bytes) {
  if ((((uintptr_t)dst | (uintptr_t)src | bytes) & 7) == 0) {
    const size_t n = bytes >> 3;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) reinterpret_cast<uint64_t*>(dst)[i] = reinterpret_cast<const uint64_t*>(src)[i];
  } else {
    for (size_t i = threadIdx.x; i < bytes; i += blockDim.x) dst[i] = src[i];
  }
}

__global__ void __launch_bounds__(256) k_hr_post(const char* src, char* stage, size_t bytes, uint64_t* posted, uint64_t seq) {
  hr_copy(stage, src, bytes);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(posted, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void __launch_bounds__(256) k_hr_fetch(char* dst, const char* stage, size_t bytes, const uint64_t* ready, uint64_t seq,
                                                  uint64_t* fetched, uint64_t* status, long long ticks) {
  __shared__ int ok;
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    int good = 0;
    for (;;) {   // bounded: the helper releases `ready` for failed operations too, this is the second line of defence
      if (__hip_atomic_load(ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= seq) { good = 1; break; }
      if (wall_clock64() - t0 > ticks) break;
      __builtin_amdgcn_s_sleep(64);
    }
    ok = good;
  }
  __syncthreads();
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  if (ok) hr_copy(dst, stage, bytes);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!ok) __hip_atomic_store(status, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(fetched, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---- the entry points --------------------------------------------------------------------------------------------
struct Pending { HrOp op; const void* src; void* dst; HrComm* c; hipStream_t st; };
thread_local int t_group = 0;
thread_local std::vector<Pending> t_pending;

void note_stream(HrComm* c, hipStream_t st) {
  for (hipStream_t s : c->streams) if (s == st) return;
  c->streams.push_back(st);
}

ncclResult_t launch(const Pending& p) {
  HrComm* c = p.c;
  const uint64_t k = c->enq;
  if (k >= HR_NS) {   // back-pressure: the slot's previous operation must be through the helper (and the stream)
    const uint64_t j = k - HR_NS;
    const bool in = c->ring[j % HR_NS].has_in;
    if (!spin_until(c, [&] { return c->helper_done.load(std::memory_order_acquire) > j && (!in || ld_acq(c->flags + F_FETCHED) >= j + 1 || c->failed.load()); },
                    "waiting for a free staging slot"))
      return ncclSystemError;
  }
  c->ring[k % HR_NS] = p.op;
  note_stream(c, p.st);
  if (p.op.kind != OP_RECV)
    hipLaunchKernelGGL(k_hr_post, dim3(1), dim3(256), 0, p.st, (const char*)p.src, c->stage_out + (k % HR_NS) * HR_STAGE_CAP, p.op.bytes,
                       c->flags + F_POSTED, k + 1);
  {
    std::lock_guard<std::mutex> lk(c->mu);
    c->enq_pub = k + 1;
  }
  c->cv.notify_one();
  if (p.op.has_in)
    hipLaunchKernelGGL(k_hr_fetch, dim3(1), dim3(256), 0, p.st, (char*)p.dst, c->stage_in + (k % HR_NS) * HR_STAGE_CAP, p.op.bytes,
                       c->flags + F_READY, k + 1, c->flags + F_FETCHED, c->flags + F_STATUS, (long long)(2.0 * c->timeout_s * 1e8));
  c->enq = k + 1;
  return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t submit(const Pending& p) {
  if (!p.c || p.c->abort.load()) return ncclInvalidUsage;
  if (p.c->failed.load()) return ncclSystemError;   // like RCCL after an asynchronous error: nothing more is accepted
  if (p.op.bytes == 0 || p.op.bytes > (p.op.kind <= OP_BCAST ? HR_AR_CAP : HR_P2P_CAP)) {
    fprintf(stderr, "[pyapes hostring] operation of %zu bytes outside the stand-in's capacity\n", p.op.bytes);
    return ncclInvalidArgument;
  }
  if (t_group > 0) { t_pending.push_back(p); return ncclSuccess; }
  return launch(p);
}

ncclResult_t hr_GetUniqueId(ncclUniqueId* id) {
  static std::atomic<int> ctr{0};
  memset(id, 0, sizeof(*id));
  unsigned long long r = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
  snprintf(id->internal, sizeof(id->internal), "/pyapes_hr_%d_%d_%llx", (int)getpid(), ctr.fetch_add(1), r);
  int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return ncclSystemError;
  const int rc = ftruncate(fd, (off_t)HR_BYTES);   // zero-filled and sparse: all counters start at 0
  close(fd);
  return rc == 0 ? ncclSuccess : ncclSystemError;
}

void hr_free(HrComm* c) {
  if (c->stage_out) (void)hipHostFree(c->stage_out);
  if (c->stage_in) (void)hipHostFree(c->stage_in);
  if (c->flags) (void)hipHostFree(c->flags);
  if (c->shm) munmap(c->shm, HR_BYTES);
  delete c;
}

ncclResult_t hr_CommInitRank(ncclComm_t* out, int n, ncclUniqueId id, int rank) {
  const int call = g_init_calls++;
  if (n < 1 || n > HR_MAXR || rank < 0 || rank >= n) return ncclInvalidArgument;
  id.internal[sizeof(id.internal) - 1] = 0;
  int fd = shm_open(id.internal, O_RDWR, 0600);
  if (fd < 0) return ncclSystemError;
  void* m = mmap(nullptr, HR_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return ncclSystemError;
  HrComm* c = new HrComm;
  c->rank = rank;
  c->n = n;
  c->index = g_comms_made;
  c->shm = reinterpret_cast<HrShm*>(m);
  c->ar_data = reinterpret_cast<char*>(m) + HR_HDR;
  c->p2p_data = c->ar_data + HR_AR_BYTES;
  if (const char* t = getenv("PYAPES_HIP_HOSTRING_TIMEOUT")) c->timeout_s = atof(t);
  int fail_rank = -1;
  char what[16] = {0};
  long idx = 0;
  bool fail_init = false;
  if (const char* f = getenv("PYAPES_HIP_HOSTRING_FAIL")) {
    if (sscanf(f, "%d:%15[a-z]:%ld", &fail_rank, what, &idx) == 3 && fail_rank == rank) {
      if (!strcmp(what, "init")) fail_init = idx == call;
      if (!strcmp(what, "corrupt")) { c->fail_what = 2; c->fail_index = idx; }
      if (!strcmp(what, "hang")) { c->fail_what = 3; c->fail_index = idx; }
    }
  }
  // the ranks meet (what ncclCommInitRank's bootstrap does); the segment's name can go once everybody holds it
  c->shm->attached.fetch_add(1, std::memory_order_acq_rel);
  const bool met = spin_until(c, [&] { return c->shm->attached.load(std::memory_order_acquire) >= (uint32_t)n; }, "waiting for the other ranks in ncclCommInitRank");
  if (rank == 0) shm_unlink(id.internal);
  if (!met || fail_init) {
    if (fail_init) fprintf(stderr, "[pyapes hostring] rank %d: injected failure of ncclCommInitRank call %d\n", rank, call);
    hr_free(c);
    return ncclSystemError;
  }
  if (hipHostMalloc((void**)&c->stage_out, HR_NS * HR_STAGE_CAP, hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void**)&c->stage_in, HR_NS * HR_STAGE_CAP, hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void**)&c->flags, F_WORDS * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    hr_free(c);
    return ncclUnhandledCudaError;
  }
  memset(c->flags, 0, F_WORDS * sizeof(uint64_t));
  c->helper = std::thread(helper_main, c);
  ++g_comms_made;
  *out = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t hr_shutdown(ncclComm_t comm, bool abort) {
  HrComm* c = reinterpret_cast<HrComm*>(comm);
  if (!c) return ncclInvalidArgument;
  if (abort) c->abort.store(1);   // every wait of the helper and of this thread ends; queued fetch kernels are released
  {
    std::lock_guard<std::mutex> lk(c->mu);
    c->stop = true;               // the helper leaves once it has gone through what was enqueued
  }
  c->cv.notify_one();
  if (c->helper.joinable()) c->helper.join();
  for (hipStream_t s : c->streams) (void)hipStreamSynchronize(s);   // no kernel may still touch the pinned slots
  const bool bad = c->failed.load() != 0 || (c->flags && ld_acq(c->flags + F_STATUS) != 0);
  if (bad && !abort) fprintf(stderr, "[pyapes hostring] rank %d comm %d: destroyed after a failed operation\n", c->rank, c->index);
  hr_free(c);
  return ncclSuccess;
}

ncclResult_t hr_CommDestroy(ncclComm_t comm) { return hr_shutdown(comm, false); }
ncclResult_t hr_CommAbort(ncclComm_t comm) { return hr_shutdown(comm, true); }

ncclResult_t hr_CommCount(const ncclComm_t comm, int* n) {
  if (!comm || !n) return ncclInvalidArgument;
  *n = reinterpret_cast<const HrComm*>(comm)->n;
  return ncclSuccess;
}

ncclResult_t hr_AllReduce(const void* s, void* r, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t st) {
  if (op != ncclSum && op != ncclMin) return ncclInvalidArgument;
  Pending p{};
  p.op.kind = OP_AR; p.op.count = count; p.op.bytes = count * dt_size(dt); p.op.dt = dt; p.op.op = op; p.op.has_in = true;
  p.src = s; p.dst = r; p.c = reinterpret_cast<HrComm*>(comm); p.st = st;
  return submit(p);
}

ncclResult_t hr_Broadcast(const void* s, void* r, size_t count, ncclDataType_t dt, int root, ncclComm_t comm, hipStream_t st) {
  Pending p{};
  p.op.kind = OP_BCAST; p.op.count = count; p.op.bytes = count * dt_size(dt); p.op.dt = dt; p.op.root = root; p.op.has_in = true;
  p.src = s; p.dst = r; p.c = reinterpret_cast<HrComm*>(comm); p.st = st;
  return submit(p);
}

ncclResult_t hr_Send(const void* s, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) {
  Pending p{};
  p.op.kind = OP_SEND; p.op.count = count; p.op.bytes = count * dt_size(dt); p.op.dt = dt; p.op.peer = peer;
  p.src = s; p.c = reinterpret_cast<HrComm*>(comm); p.st = st;
  if (!p.c || peer < 0 || peer >= p.c->n) return ncclInvalidArgument;
  return submit(p);
}

ncclResult_t hr_Recv(void* r, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) {
  Pending p{};
  p.op.kind = OP_RECV; p.op.count = count; p.op.bytes = count * dt_size(dt); p.op.dt = dt; p.op.peer = peer; p.op.has_in = true;
  p.dst = r; p.c = reinterpret_cast<HrComm*>(comm); p.st = st;
  if (!p.c || peer < 0 || peer >= p.c->n) return ncclInvalidArgument;
  return submit(p);
}

ncclResult_t hr_GroupStart() { ++t_group; return ncclSuccess; }

ncclResult_t hr_GroupEnd() {
  if (t_group <= 0) return ncclInvalidUsage;
  if (--t_group > 0) return ncclSuccess;
  // a group's operations progress together in RCCL; here they are serial on the stream, so every send goes first
  // (a send only needs room in a 2-deep FIFO, a receive needs the peer's send)
  ncclResult_t rc = ncclSuccess;
  for (int pass = 0; pass < 2; ++pass)
    for (const Pending& p : t_pending)
      if ((p.op.kind == OP_RECV) == (pass == 1) && rc == ncclSuccess) rc = launch(p);
  t_pending.clear();
  return rc;
}

const char* hr_GetErrorString(ncclResult_t e) {
  switch (e) {
    case ncclSuccess: return "success";
    case ncclUnhandledCudaError: return "HIP error (hostring stand-in)";
    case ncclSystemError: return "system error / peer did not arrive (hostring stand-in)";
    case ncclInvalidArgument: return "invalid argument (hostring stand-in)";
    case ncclInvalidUsage: return "invalid usage (hostring stand-in)";
    default: return "error (hostring stand-in)";
  }
}

}  // namespace

// the entry points under RCCL's own names (C linkage, as rccl.h declares them): pa_comm_use_impl() dlopens this library
// RTLD_LOCAL and takes them with dlsym from ITS handle, so a librccl that is also mapped into the process is not touched
extern "C" {
__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { return hr_GetUniqueId(id); }
__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t* comm, int n, ncclUniqueId id, int rank) { return hr_CommInitRank(comm, n, id, rank); }
__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm) { return hr_CommDestroy(comm); }
__attribute__((visibility("default"))) ncclResult_t ncclCommAbort(ncclComm_t comm) { return hr_CommAbort(comm); }
__attribute__((visibility("default"))) ncclResult_t ncclCommCount(const ncclComm_t comm, int* n) { return hr_CommCount(comm, n); }
__attribute__((visibility("default"))) ncclResult_t ncclAllReduce(const void* s, void* r, size_t n, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t st) { return hr_AllReduce(s, r, n, t, op, comm, st); }
__attribute__((visibility("default"))) ncclResult_t ncclBroadcast(const void* s, void* r, size_t n, ncclDataType_t t, int root, ncclComm_t comm, hipStream_t st) { return hr_Broadcast(s, r, n, t, root, comm, st); }
__attribute__((visibility("default"))) ncclResult_t ncclSend(const void* s, size_t n, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) { return hr_Send(s, n, t, peer, comm, st); }
__attribute__((visibility("default"))) ncclResult_t ncclRecv(void* r, size_t n, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) { return hr_Recv(r, n, t, peer, comm, st); }
__attribute__((visibility("default"))) ncclResult_t ncclGroupStart() { return hr_GroupStart(); }
__attribute__((visibility("default"))) ncclResult_t ncclGroupEnd() { return hr_GroupEnd(); }
__attribute__((visibility("default"))) const char* ncclGetErrorString(ncclResult_t e) { return hr_GetErrorString(e); }
// what pa_comm_impl() reports for a library that exports it
__attribute__((visibility("default"))) const char* pa_comm_impl_name() { return "hostring (test stand-in: host shared memory, ranks may share a GPU)"; }
}
