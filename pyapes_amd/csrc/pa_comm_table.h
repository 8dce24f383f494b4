// pa_comm_table.h -- the handful of RCCL entry points pa_comm.hip uses, as a table of function pointers.
// Filled from librccl (dlopen, pa_comm.hip) in the product; tests that must run MORE THAN ONE rank on ONE GPU
// (RCCL refuses duplicate devices) select the stand-in of pa_comm_hostring.hip with the explicit hook
// PYAPES_HIP_COMM_IMPL=hostring -- same signatures, same stream-ordered semantics, host shared memory as the link.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

struct Rccl {
  void* h = nullptr;
  int tried = 0;
  const char* impl = "rccl";
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

// pa_comm_hostring.hip: fill the table with the test stand-in (never selected implicitly)
void pa_hostring_table(Rccl* R);
