// Test infrastructure -- NOT part of the product.  <rccl/rccl.h> for the host layer built for the emulated wavefront
// (csrc/host/DataParallel.cpp compiles unchanged against it).  A communicator of one rank is the identity; a communicator of N ranks
// is N PROCESSES on this machine that meet in a POSIX shared-memory segment named after the unique id: every collective copies the
// rank's buffer into its slot, waits for the others (a sense-reversing barrier on two atomics in the segment), reduces the slots IN RANK
// ORDER -- so every rank computes the same bits -- and waits again before the slots are reused.  Calls are synchronous and carried out in
// program order (the stream argument is dropped, ncclGroupStart / End are brackets without effect): ranks that issue different sequences
// of collectives hang, which is the property tests/test_wave_emul_cpu.py's two-rank run is there to check of ExpRunner::TrainStep.
// ncclAvg: the sum of the ranks' values in fp32, divided by the rank count, rounded once to the buffer's type (RCCL pre-multiplies by
// 1/N and sums in the buffer's type, in ring order: values may differ in the last bit, replicas are identical either way).
#pragma once
#include <fcntl.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <algorithm>
#include <string>
#include <vector>

#include <c10/util/Half.h>

typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclSystemError = 2, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclChar = 0, ncclInt32 = 2, ncclInt = 2, ncclInt64 = 4, ncclHalf = 6, ncclFloat = 7 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclMax = 2, ncclAvg = 4 } ncclRedOp_t;

namespace wemu_nccl {
constexpr size_t kSlot = 8u << 20;  // bytes of one rank's slot: larger buffers travel in pieces
struct Header {
  std::atomic<int> arrived, sense, attached;
};
inline size_t size_of(ncclDataType_t t) { return t == ncclChar ? 1 : t == ncclHalf ? 2 : t == ncclInt64 ? 8 : 4; }
}  // namespace wemu_nccl

struct ncclComm {
  int world = 1, rank = 0, sense = 0;
  wemu_nccl::Header* hdr = nullptr;
  char* slots = nullptr;
  size_t bytes = 0;
  std::string name;
  void barrier() {
    sense ^= 1;
    if (hdr->arrived.fetch_add(1) == world - 1) {
      hdr->arrived.store(0);
      hdr->sense.store(sense);
    } else {
      const timespec nap = {0, 20000};
      while (hdr->sense.load() != sense) nanosleep(&nap, nullptr);
    }
  }
  char* slot(int r) const { return slots + (size_t) r * wemu_nccl::kSlot; }
};

static inline const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "emulated RCCL (shared memory) failed"; }
static inline ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  timespec t;
  clock_gettime(CLOCK_REALTIME, &t);
  const uint64_t v[2] = {(uint64_t) t.tv_sec * 1000000007ull + (uint64_t) t.tv_nsec, (uint64_t) getpid()};
  memcpy(id->internal, v, sizeof(v));
  return ncclSuccess;
}
static inline ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
  if (world < 1 || rank < 0 || rank >= world) return ncclInvalidArgument;
  ncclComm* c = new ncclComm();
  c->world = world;
  c->rank = rank;
  *out = c;
  if (world == 1) return ncclSuccess;
  uint64_t v[2];
  memcpy(v, id.internal, sizeof(v));
  char name[64];
  snprintf(name, sizeof(name), "/wemu_nccl_%llx_%llx", (unsigned long long) v[0], (unsigned long long) v[1]);
  c->name = name;
  c->bytes = 4096 + (size_t) world * wemu_nccl::kSlot;
  const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t) c->bytes) != 0) return ncclSystemError;  // (a fresh segment reads as zeros: the barrier's initial state)
  void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  c->hdr = static_cast<wemu_nccl::Header*>(p);
  c->slots = static_cast<char*>(p) + 4096;
  c->hdr->attached.fetch_add(1);
  const timespec nap = {0, 200000};
  while (c->hdr->attached.load() < world) nanosleep(&nap, nullptr);  // (everybody has mapped the segment before anybody unlinks it)
  c->barrier();
  if (rank == 0) shm_unlink(name);
  return ncclSuccess;
}
static inline ncclResult_t ncclCommCount(ncclComm_t c, int* n) { *n = c->world; return ncclSuccess; }
static inline ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (c->hdr != nullptr) munmap(c->hdr, c->bytes);
  delete c;
  return ncclSuccess;
}
static inline ncclResult_t ncclGroupStart() { return ncclSuccess; }
static inline ncclResult_t ncclGroupEnd() { return ncclSuccess; }

namespace wemu_nccl {
template <typename T, typename Acc>
inline void reduce(const ncclComm& c, size_t n, ncclRedOp_t op, T* out) {
  for (size_t i = 0; i < n; i++) {
    Acc a = (Acc) reinterpret_cast<const T*>(c.slot(0))[i];
    for (int r = 1; r < c.world; r++) {
      const Acc b = (Acc) reinterpret_cast<const T*>(c.slot(r))[i];
      a = op == ncclMax ? std::max(a, b) : a + b;
    }
    if (op == ncclAvg) a = a / (Acc) c.world;
    out[i] = (T) a;
  }
}
}  // namespace wemu_nccl

template <typename S>
static inline ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, S) {
  const size_t es = wemu_nccl::size_of(t);
  if (c->world == 1) {
    if (send != recv) memmove(recv, send, count * es);
    return ncclSuccess;
  }
  const size_t per = wemu_nccl::kSlot / es;
  for (size_t off = 0; off < count || (count == 0 && off == 0); off += per) {  // (a zero-length collective still meets the others)
    const size_t n = std::min(per, count - off);
    memcpy(c->slot(c->rank), static_cast<const char*>(send) + off * es, n * es);
    c->barrier();
    char* out = static_cast<char*>(recv) + off * es;
    if (t == ncclFloat) wemu_nccl::reduce<float, float>(*c, n, op, reinterpret_cast<float*>(out));
    else if (t == ncclHalf) wemu_nccl::reduce<c10::Half, float>(*c, n, op, reinterpret_cast<c10::Half*>(out));
    else if (t == ncclInt32) wemu_nccl::reduce<int32_t, int64_t>(*c, n, op, reinterpret_cast<int32_t*>(out));
    else if (t == ncclInt64) wemu_nccl::reduce<int64_t, int64_t>(*c, n, op, reinterpret_cast<int64_t*>(out));
    else return ncclInvalidArgument;
    c->barrier();
    if (count == 0) break;
  }
  return ncclSuccess;
}
template <typename S>
static inline ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, S) {
  const size_t bytes = count * wemu_nccl::size_of(t);
  if (c->world == 1) {
    if (send != recv) memmove(recv, send, bytes);
    return ncclSuccess;
  }
  for (size_t off = 0; off < bytes || (bytes == 0 && off == 0); off += wemu_nccl::kSlot) {
    const size_t n = std::min(wemu_nccl::kSlot, bytes - off);
    if (c->rank == root) memcpy(c->slot(root), static_cast<const char*>(send) + off, n);
    c->barrier();
    if (c->rank != root || send != recv) memcpy(static_cast<char*>(recv) + off, c->slot(root), n);
    c->barrier();
    if (bytes == 0) break;
  }
  return ncclSuccess;
}
