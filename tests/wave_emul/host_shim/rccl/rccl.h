// Test infrastructure -- NOT part of the product.  <rccl/rccl.h> for the host layer built for the emulated wavefront: a world of
// ONE rank, in which every collective is the identity (csrc/host/DataParallel.cpp compiles unchanged; its one-rank hooks run).
#pragma once
#include <stddef.h>
#include <string.h>
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclChar = 0, ncclInt32 = 2, ncclInt = 2, ncclInt64 = 4, ncclHalf = 6, ncclFloat = 7 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclMax = 2, ncclAvg = 4 } ncclRedOp_t;
static inline const char* ncclGetErrorString(ncclResult_t) { return "emulated one-rank world"; }
static inline ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { memset(id, 7, sizeof(*id)); return ncclSuccess; }
static inline ncclResult_t ncclCommInitRank(ncclComm_t* c, int world, ncclUniqueId, int rank) {
  *c = (ncclComm_t) 0x1;
  return world == 1 && rank == 0 ? ncclSuccess : ncclInvalidArgument;
}
static inline ncclResult_t ncclCommCount(ncclComm_t, int* n) { *n = 1; return ncclSuccess; }
static inline ncclResult_t ncclCommDestroy(ncclComm_t) { return ncclSuccess; }
static inline ncclResult_t ncclGroupStart() { return ncclSuccess; }
static inline ncclResult_t ncclGroupEnd() { return ncclSuccess; }
static inline size_t wemu_nccl_size(ncclDataType_t t) { return t == ncclChar ? 1 : t == ncclHalf ? 2 : t == ncclInt64 ? 8 : 4; }
template <typename S>
static inline ncclResult_t ncclAllReduce(const void* s, void* d, size_t n, ncclDataType_t t, ncclRedOp_t, ncclComm_t, S) {
  if (s != d) memmove(d, s, n * wemu_nccl_size(t));
  return ncclSuccess;
}
template <typename S>
static inline ncclResult_t ncclBroadcast(const void* s, void* d, size_t n, ncclDataType_t t, int, ncclComm_t, S) {
  if (s != d) memmove(d, s, n * wemu_nccl_size(t));
  return ncclSuccess;
}
