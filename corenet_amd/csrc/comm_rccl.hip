// RCCL driven from the library (the reference reaches NCCL through torch's DistributedDataParallel,
// pipeline.py:199-200,229): communicator bootstrap from a 128-byte unique id that the host passes around (the Python
// side broadcasts it through the torch.distributed store it already has), and an in-place fp32 all-reduce enqueued on
// the CALLER'S stream -- the engine's side stream, right behind the un-pack of a gradient bucket, so the exchange of a
// bucket needs no extra stream hop and no host round trip.
//
// librccl.so is opened lazily (dlopen): the library itself loads -- and every other entry point works -- on a box
// without RCCL; the crn_comm_* calls then return CRN_EINVAL.
#include "crn_common.h"
#include <dlfcn.h>
#include <cstdio>
#include <cstring>

namespace {
// the part of rccl.h this file uses (RCCL keeps NCCL's ABI)
struct NcclUniqueId { char internal[128]; };
typedef void* NcclComm;
typedef int (*GetUniqueIdFn)(NcclUniqueId*);
typedef int (*CommInitRankFn)(NcclComm*, int, NcclUniqueId, int);
typedef int (*CommDestroyFn)(NcclComm);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
typedef int (*BroadcastFn)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
typedef int (*GetVersionFn)(int*);
typedef const char* (*GetErrorStringFn)(int);
constexpr int kNcclFloat32 = 7, kNcclSum = 0;          // ncclDataType_t / ncclRedOp_t values (rccl.h)

struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_id = nullptr;
  CommInitRankFn init_rank = nullptr;
  CommDestroyFn destroy = nullptr;
  AllReduceFn all_reduce = nullptr;
  BroadcastFn broadcast = nullptr;
  GetVersionFn version = nullptr;
  GetErrorStringFn errstr = nullptr;
  bool ok = false;
};
Rccl& rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) if ((r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
    if (r.handle) {
      r.get_id = (GetUniqueIdFn)dlsym(r.handle, "ncclGetUniqueId");
      r.init_rank = (CommInitRankFn)dlsym(r.handle, "ncclCommInitRank");
      r.destroy = (CommDestroyFn)dlsym(r.handle, "ncclCommDestroy");
      r.all_reduce = (AllReduceFn)dlsym(r.handle, "ncclAllReduce");
      r.broadcast = (BroadcastFn)dlsym(r.handle, "ncclBroadcast");
      r.version = (GetVersionFn)dlsym(r.handle, "ncclGetVersion");
      r.errstr = (GetErrorStringFn)dlsym(r.handle, "ncclGetErrorString");
      r.ok = r.get_id && r.init_rank && r.destroy && r.all_reduce && r.broadcast && r.version;
    }
  }
  return r;
}
int fail(int rc, const char* what) {
  Rccl& r = rccl();
  fprintf(stderr, "[corenet_hip] %s: RCCL error %d (%s)\n", what, rc, r.errstr ? r.errstr(rc) : "?");
  return 1000 + rc;           // positive: a runtime error, like the hipError_t codes
}
struct CrnComm { NcclComm comm; int rank, nranks; };
}  // namespace

extern "C" int crn_comm_unique_id(void* id128) {
  Rccl& r = rccl();
  if (!r.ok || !id128) return CRN_EINVAL;
  NcclUniqueId id;
  const int rc = r.get_id(&id);
  if (rc) return fail(rc, "ncclGetUniqueId");
  memcpy(id128, &id, sizeof(id));
  return CRN_OK;
}

extern "C" int crn_comm_init(const void* id128, int rank, int nranks, void** comm) {
  Rccl& r = rccl();
  if (!r.ok || !id128 || !comm || nranks < 1 || rank < 0 || rank >= nranks) return CRN_EINVAL;
  NcclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  CrnComm* c = new CrnComm{nullptr, rank, nranks};
  const int rc = r.init_rank(&c->comm, nranks, id, rank);       // (the current HIP device becomes the communicator's)
  if (rc) { delete c; return fail(rc, "ncclCommInitRank"); }
  *comm = c;
  return CRN_OK;
}

extern "C" int crn_comm_destroy(void* comm) {
  Rccl& r = rccl();
  if (!r.ok || !comm) return CRN_EINVAL;
  CrnComm* c = reinterpret_cast<CrnComm*>(comm);
  const int rc = r.destroy(c->comm);
  delete c;
  return rc ? fail(rc, "ncclCommDestroy") : CRN_OK;
}

extern "C" int crn_comm_info(void* comm, int* rccl_version, int* rank, int* nranks) {
  Rccl& r = rccl();
  if (!r.ok) return CRN_EINVAL;
  if (rccl_version) { const int rc = r.version(rccl_version); if (rc) return fail(rc, "ncclGetVersion"); }
  if (comm) {
    CrnComm* c = reinterpret_cast<CrnComm*>(comm);
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
  }
  return CRN_OK;
}

extern "C" int crn_allreduce_f32(void* comm, float* buf, int64_t n, crnStream stream) {
  CRN_ENTRY(stream);
  Rccl& r = rccl();
  if (!r.ok || !comm || !buf || n < 0) return CRN_EINVAL;
  if (n == 0) return CRN_OK;
  CrnComm* c = reinterpret_cast<CrnComm*>(comm);
  const int rc = r.all_reduce(buf, buf, (size_t)n, kNcclFloat32, kNcclSum, c->comm, (hipStream_t)stream);
  return rc ? fail(rc, "ncclAllReduce") : CRN_OK;
}

extern "C" int crn_broadcast_f32(void* comm, float* buf, int64_t n, int root, crnStream stream) {
  CRN_ENTRY(stream);
  Rccl& r = rccl();
  if (!r.ok || !comm || !buf || n < 0) return CRN_EINVAL;
  if (n == 0) return CRN_OK;
  CrnComm* c = reinterpret_cast<CrnComm*>(comm);
  const int rc = r.broadcast(buf, buf, (size_t)n, kNcclFloat32, root, c->comm, (hipStream_t)stream);
  return rc ? fail(rc, "ncclBroadcast") : CRN_OK;
}

// ---------------- rocprofv3 markers --------------------------------------------------------------------------------
// roctx ranges around the library calls of one layer (the reference has no tracing hooks at all):
// with CRN_ROCTX=1 the Python engine brackets every layer's launches with crn_roctx_push(label) / crn_roctx_pop(), and
//   rocprofv3 --kernel-trace --marker-trace -- python ...
// attributes kernels to layers without the event-probe tools.  The marker library is opened on first use
// (librocprofiler-sdk-roctx.so, else libroctx64.so); without it the calls are no-ops that return CRN_EINVAL.
namespace {
typedef int (*RoctxPushFn)(const char*);
typedef int (*RoctxPopFn)();
struct Roctx { RoctxPushFn push = nullptr; RoctxPopFn pop = nullptr; };
Roctx& roctx() {
  static Roctx r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char* names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
    for (const char* n : names) {
      void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      r.push = (RoctxPushFn)dlsym(h, "roctxRangePushA");
      r.pop = (RoctxPopFn)dlsym(h, "roctxRangePop");
      if (r.push && r.pop) break;
      r.push = nullptr; r.pop = nullptr;
    }
  }
  return r;
}
}  // namespace
extern "C" int crn_roctx_push(const char* label) {
  Roctx& r = roctx();
  if (!r.push || !label) return CRN_EINVAL;
  r.push(label);
  return CRN_OK;
}
extern "C" int crn_roctx_pop(void) {
  Roctx& r = roctx();
  if (!r.pop) return CRN_EINVAL;
  r.pop();
  return CRN_OK;
}
