// RCCL helpers of the C ABI (comm_init from a unique id, allreduce_bucket): the data-parallel exchange of
// the training step -- one sum all-reduce per gradient bucket of the flat f32 gradient buffer, on the caller's stream --
// for hosts that drive libctdet_hip.so without torch.distributed.  Replaces what DistributedDataParallel's reducer does
// over NCCL in the reference (detectron2/engine/defaults.py:279-285).  librccl is opened on first use (dlopen), so the
// library itself loads on machines without it.
#include "common.h"
#include "../../include/ctdet_hip.h"
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>
#include <mutex>

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::once_flag g_once;

bool rccl_load() {
  std::call_once(g_once, [] {
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    Rccl r;
    r.h = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.Broadcast = (decltype(r.Broadcast))dlsym(h, "ncclBroadcast");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (r.GetUniqueId && r.CommInitRank && r.AllReduce && r.Broadcast && r.CommDestroy && r.GetErrorString) g_rccl = r;
  });
  return g_rccl.h != nullptr;
}
}  // namespace

#define RCCL_CALL(expr, what)                                                           \
  do {                                                                                  \
    ncclResult_t r_ = (expr);                                                           \
    if (r_ != ncclSuccess) {                                                            \
      ctdet_set_error("%s failed: %s", what, g_rccl.GetErrorString(r_));               \
      return -5;                                                                        \
    }                                                                                   \
  } while (0)

extern "C" {

int32_t ctdet_comm_unique_id(void* id_out) {
  CTDET_CHECK(id_out, "comm_unique_id: null pointer");
  CTDET_CHECK(rccl_load(), "comm: librccl.so could not be loaded (%s)", dlerror() ? dlerror() : "symbols missing");
  static_assert(sizeof(ncclUniqueId) == CTDET_COMM_ID_BYTES, "unique id size");
  RCCL_CALL(g_rccl.GetUniqueId((ncclUniqueId*)id_out), "ncclGetUniqueId");
  return 0;
}

int32_t ctdet_comm_init(const void* id, int32_t rank, int32_t world, void** comm_out) {
  CTDET_CHECK(id && comm_out, "comm_init: null pointer");
  CTDET_CHECK(world >= 1 && rank >= 0 && rank < world, "comm_init: bad rank %d of %d", rank, world);
  CTDET_CHECK(rccl_load(), "comm: librccl.so could not be loaded");
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t c = nullptr;
  RCCL_CALL(g_rccl.CommInitRank(&c, world, uid, rank), "ncclCommInitRank");   // binds to the calling thread's current device
  *comm_out = (void*)c;
  return 0;
}

int32_t ctdet_allreduce_bucket(void* comm, float* buf, int64_t count, void* stream) {
  CTDET_CHECK(comm && (buf || count == 0), "allreduce_bucket: null pointer");
  if (count == 0) return 0;
  RCCL_CALL(g_rccl.AllReduce(buf, buf, (size_t)count, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream),
            "ncclAllReduce");
  return 0;
}

int32_t ctdet_bcast(void* comm, float* buf, int64_t count, int32_t root, void* stream) {
  CTDET_CHECK(comm && (buf || count == 0), "bcast: null pointer");
  if (count == 0) return 0;
  RCCL_CALL(g_rccl.Broadcast(buf, buf, (size_t)count, ncclFloat32, root, (ncclComm_t)comm, (hipStream_t)stream),
            "ncclBroadcast");
  return 0;
}

int32_t ctdet_comm_destroy(void* comm) {
  if (!comm) return 0;
  CTDET_CHECK(rccl_load(), "comm: librccl.so could not be loaded");
  RCCL_CALL(g_rccl.CommDestroy((ncclComm_t)comm), "ncclCommDestroy");
  return 0;
}

}  // extern "C"
