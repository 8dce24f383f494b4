// pa_comm_table.h -- the handful of RCCL entry points pa_comm.hip uses, as a table of function pointers.
// Filled from librccl (dlopen, pa_comm.hip) -- or from the library a caller names through pa_comm_use_impl(path) before
// the first communicator exists: tests that must run MORE THAN ONE rank on ONE GPU (RCCL refuses duplicate devices)
// hand in tests/lib/libpa_hostring.so, which exports the same twelve symbols with the same stream-ordered semantics over
// host shared memory.  Nothing of that stand-in is linked into this library.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

struct Rccl {
  void* h = nullptr;
  int tried = 0;
  const char* impl = "rccl";
  char impl_buf[256] = "";
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
