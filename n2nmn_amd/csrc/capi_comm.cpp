// Gradient all-reduce over RCCL inside the C-ABI (SURVEY.md section 8(b) item 6, 8(e) "Collective").
//
// The reference trains on one GPU; the data-parallel step of this build all-reduces ONE flat fp32
// gradient vector in two buckets (train.py): the late part [split, numel) -- decoder + module
// variables -- is final after backward phase 0 and travels while the encoder's BPTT (phase 1) runs,
// the early part [0, split) follows.  Here the collective is issued by the library itself on a side
// stream it owns, forked from and joined to the caller's stream with events, so a host program needs
// no PyTorch process group for the data path (torch.distributed -- or any other channel -- is only
// needed to hand the 128-byte ncclUniqueId from rank 0 to the other ranks).
//
// librccl is bound at run time (dlopen): the library keeps loading on hosts without RCCL, and inside
// a PyTorch process it attaches to the librccl PyTorch already mapped instead of a second copy.
#include <dlfcn.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// Build hosts without the RCCL headers: the handful of ABI-stable types and constants this file
// needs (every function is bound with dlsym at run time anyway).
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat32 = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
#endif

#include <cstring>
#include <mutex>

#include "ctx.h"

namespace {

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)                        // a copy that is already mapped wins
      if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names)
      if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!r.h) { r.error = std::string("librccl not found: ") + dlerror(); return; }
    auto sym = [&](const char* s) {
      void* p = dlsym(r.h, s);
      if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + s;
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return r;
}

#define N2_NCCL(expr)                                                                       \
  do {                                                                                      \
    ncclResult_t _r = (expr);                                                               \
    if (_r != ncclSuccess) {                                                                \
      set_last_error(std::string(#expr) + ": " +                                            \
                     (rccl().GetErrorString ? rccl().GetErrorString(_r) : "rccl error"));  \
      return N2NMN_EHIP;                                                                    \
    }                                                                                       \
  } while (0)

}  // namespace

struct n2nmn_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t side = nullptr;          // the collectives run here
  hipEvent_t fork = nullptr;           // caller's stream -> side
  hipEvent_t done[2] = {nullptr, nullptr};
  bool pending[2] = {false, false};
};

extern "C" {

int n2nmn_comm_unique_id(void* id_out_128) {
  N2_REQUIRE(id_out_128, N2NMN_EINVAL, "comm_unique_id: null argument");
  N2_REQUIRE(rccl().error.empty(), N2NMN_EHIP, rccl().error);
  static_assert(sizeof(ncclUniqueId) == N2NMN_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  N2_NCCL(rccl().GetUniqueId(&id));
  std::memcpy(id_out_128, &id, sizeof(id));
  return N2NMN_OK;
}

int n2nmn_comm_create(const void* unique_id_128, int rank, int world, int device, n2nmn_comm** out) {
  N2_REQUIRE(unique_id_128 && out, N2NMN_EINVAL, "comm_create: null argument");
  N2_REQUIRE(world >= 1 && rank >= 0 && rank < world, N2NMN_EINVAL, "comm_create: bad rank / world");
  N2_REQUIRE(rccl().error.empty(), N2NMN_EHIP, rccl().error);
  N2_HIP(hipSetDevice(device));
  n2nmn_comm* c = new n2nmn_comm();
  c->rank = rank; c->world = world; c->device = device;
  ncclUniqueId id;
  std::memcpy(&id, unique_id_128, sizeof(id));
  ncclResult_t r = rccl().CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    set_last_error(std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
    delete c;
    return N2NMN_EHIP;
  }
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  // any failure from here on releases what exists already (communicator, stream, events)
  hipError_t e = hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, hi);   // collectives first
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->fork, hipEventDisableTiming);
  for (auto& ev : c->done)
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  if (e != hipSuccess) {
    set_last_error(std::string("comm_create: ") + hipGetErrorString(e));
    (void)n2nmn_comm_destroy(c);
    return N2NMN_EHIP;
  }
  *out = c;
  return N2NMN_OK;
}

int n2nmn_comm_world(const n2nmn_comm* comm) { return comm ? comm->world : N2NMN_EINVAL; }

int n2nmn_allreduce_grads(n2nmn_ctx* ctx, n2nmn_comm* comm, int bucket, float* grads,
                          n2nmn_stream stream) {
  N2_REQUIRE(ctx && comm && grads, N2NMN_EINVAL, "allreduce_grads: null argument");
  N2_REQUIRE(bucket == 0 || bucket == 1, N2NMN_EINVAL, "allreduce_grads: bucket is 0 (late) or 1 (early)");
  const int64_t numel = n2nmn_grad_numel(ctx), split = n2nmn_grad_split(ctx);
  N2_REQUIRE(numel > 0, N2NMN_EINVAL, "allreduce_grads: call n2nmn_train_enable first");
  // bucket 0 = [split, numel): decoder + module variables, final after backward phase 0;
  // bucket 1 = [0, split): encoder variables, final after phase 1
  float* p = bucket == 0 ? grads + split : grads;
  const size_t n = (size_t)(bucket == 0 ? numel - split : split);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  N2_HIP(hipEventRecord(comm->fork, s));                   // the gradients of this bucket are final
  N2_HIP(hipStreamWaitEvent(comm->side, comm->fork, 0));
  n2nmn::train_side_join(ctx, comm->side);                 // (N2NMN_BWD_DEFER_JOIN: leaves still in flight)
  N2_NCCL(rccl().AllReduce(p, p, n, ncclFloat32, ncclSum, comm->comm, comm->side));
  N2_HIP(hipEventRecord(comm->done[bucket], comm->side));
  comm->pending[bucket] = true;
  return N2NMN_OK;
}

int n2nmn_allreduce_wait(n2nmn_comm* comm, n2nmn_stream stream) {
  N2_REQUIRE(comm, N2NMN_EINVAL, "allreduce_wait: null argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  for (int b = 0; b < 2; ++b) {
    if (!comm->pending[b]) continue;
    N2_HIP(hipStreamWaitEvent(s, comm->done[b], 0));       // the optimiser step is ordered after it
    comm->pending[b] = false;
  }
  return N2NMN_OK;
}

int n2nmn_comm_destroy(n2nmn_comm* comm) {
  if (!comm) return N2NMN_OK;
  if (comm->side) (void)hipStreamSynchronize(comm->side);
  if (comm->comm && rccl().CommDestroy) (void)rccl().CommDestroy(comm->comm);
  if (comm->fork) (void)hipEventDestroy(comm->fork);
  for (auto& e : comm->done)
    if (e) (void)hipEventDestroy(e);
  if (comm->side) (void)hipStreamDestroy(comm->side);
  delete comm;
  return N2NMN_OK;
}

}  // extern "C"
