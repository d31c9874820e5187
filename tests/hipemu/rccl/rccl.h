// TEST INFRASTRUCTURE ONLY -- in-process stand-in for the RCCL calls of owshen_amd/csrc/multi.hip, for the
// single-threaded CPU interpreter build (tests/hipemu).  "Devices" are heap blocks of one process, so a grouped
// collective is a set of memcpys executed at ncclGroupEnd.  The product links the real librccl.
#pragma once
#include <stddef.h>
#include <string.h>
#include <vector>

enum ncclResult_t { ncclSuccess = 0, ncclInvalidUsage = 5 };
enum ncclDataType_t { ncclUint8 = 1 };
struct emu_nccl_comm { int rank, n; };
typedef emu_nccl_comm* ncclComm_t;

namespace emu_nccl {
struct Op { int kind; const void* send; void* recv; size_t bytes; int root, rank, n; };
inline std::vector<Op>& queue() { static std::vector<Op> q; return q; }
inline int& depth() { static int d = 0; return d; }
inline ncclResult_t flush() {
  std::vector<Op> q;
  q.swap(queue());
  if (q.empty()) return ncclSuccess;
  const int n = q[0].n;
  if ((int)q.size() != n) return ncclInvalidUsage;  // every rank must take part in the group
  for (const Op& o : q)
    if (o.kind != q[0].kind || o.bytes != q[0].bytes || o.n != n) return ncclInvalidUsage;
  if (q[0].kind == 0) {  // all-gather
    for (const Op& dst : q)
      for (const Op& src : q) memcpy((char*)dst.recv + (size_t)src.rank * src.bytes, src.send, src.bytes);
  } else {  // broadcast
    const Op* root = nullptr;
    for (const Op& o : q)
      if (o.rank == o.root) root = &o;
    if (!root) return ncclInvalidUsage;
    for (const Op& dst : q)
      if (dst.recv != root->send) memcpy(dst.recv, root->send, root->bytes);
  }
  return ncclSuccess;
}
}  // namespace emu_nccl

static inline const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ncclSuccess(emu)" : "ncclInvalidUsage(emu)"; }
static inline ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int*) {
  for (int r = 0; r < n; r++) comms[r] = new emu_nccl_comm{r, n};
  return ncclSuccess;
}
static inline ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return ncclSuccess; }
static inline ncclResult_t ncclCommCount(ncclComm_t c, int* n) { *n = c->n; return ncclSuccess; }
static inline ncclResult_t ncclCommUserRank(ncclComm_t c, int* r) { *r = c->rank; return ncclSuccess; }
static inline ncclResult_t ncclCommCuDevice(ncclComm_t c, int* d) { *d = c->rank; return ncclSuccess; }
static inline ncclResult_t ncclGroupStart() { emu_nccl::depth()++; return ncclSuccess; }
static inline ncclResult_t ncclGroupEnd() { return --emu_nccl::depth() == 0 ? emu_nccl::flush() : ncclSuccess; }
static inline ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t, ncclComm_t c, void*) {
  emu_nccl::queue().push_back({0, send, recv, count, 0, c->rank, c->n});
  return emu_nccl::depth() ? ncclSuccess : emu_nccl::flush();
}
static inline ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t, int root, ncclComm_t c, void*) {
  emu_nccl::queue().push_back({1, send, recv, count, root, c->rank, c->n});
  return emu_nccl::depth() ? ncclSuccess : emu_nccl::flush();
}
