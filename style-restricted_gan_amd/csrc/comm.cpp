// Gradient / statistics collectives of the data-parallel step over RCCL (xGMI), behind the C ABI (include/srgan_hip.h,
// "collectives"): what replaces nn.DataParallel's scatter / gather of notebook/05-train_Style-Restricted_GAN.ipynb:404-407,446.
//
// librccl.so is bound LAZILY (dlopen at the first call that needs it): libsrgan_hip.so keeps loading on a host without RCCL or
// without a GPU (tests/test_abi_cpu.py), and a single-GPU run never maps the 300 MB library.  The entry points only enqueue: an
// all-reduce is one RCCL call on the caller's stream, in place on a caller-owned device buffer, so it can sit inside a hipGraph
// capture on a communication stream (RCCL collectives are capturable) as well as run eagerly.  Nothing here allocates device
// memory, synchronises or spawns threads; rendezvous (who is rank r of n, the 128-byte unique id) is the caller's business --
// the Python host moves the id over its torch.distributed control group (srgan_amd/dp.py).
#include <dlfcn.h>
#include <mutex>
#include "common.h"

namespace srgan {
namespace {

// the slice of rccl.h this file needs (RCCL keeps NCCL's ABI: enums and the 128-byte id are part of it)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                                   // ncclSuccess = 0
enum { kNcclSum = 0, kNcclAvg = 4 };                        // ncclRedOp_t
enum { kNcclFloat32 = 7, kNcclBfloat16 = 9 };               // ncclDataType_t

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  int version = 0;                                          // ncclGetVersion: major * 10000 + minor * 100 + patch
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // first the copy the process has ALREADY mapped (torch.distributed's nccl backend links librccl: a second copy
    // of the library would be a second set of communicator state beside it), only then a fresh load
    static const char* const names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* name : names) {
      r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
      if (r.handle) break;
    }
    for (const char* name : names) {
      if (r.handle) break;
      r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    }
    if (!r.handle) return;
    auto sym = [&](const char* n) { return dlsym(r.handle, n); };
    // the hand-copied ABI slice above is NCCL 2.x's (enum values, the by-value 128-byte id): refuse any other major version
    auto get_version = reinterpret_cast<ncclResult_t (*)(int*)>(sym("ncclGetVersion"));
    int version = 0;
    if (!get_version || get_version(&version) != 0 || version < 20000 || version >= 30000) { r.version = version; return; }
    r.version = version;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.CommCount && r.AllReduce && r.AllGather && r.GetErrorString;
  });
  return r;
}

int need_rccl(const char* what) {
  if (rccl().ok) return 0;
  set_error("%s: librccl.so could not be loaded (%s, ncclGetVersion %d)", what,
            rccl().handle ? "missing symbols or an ABI this file was not written against" : "dlopen failed", rccl().version);
  return -1;
}

int check_nccl(ncclResult_t e, const char* what) {
  if (e == 0) return 0;
  set_error("%s: %s", what, rccl().GetErrorString(e));
  return 1000 + e;
}

}  // namespace
}  // namespace srgan

using namespace srgan;

extern "C" int srgan_comm_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int srgan_comm_unique_id(void* id128) {
  SRGAN_REQUIRE(id128, "srgan_comm_unique_id: null output");
  if (int e = need_rccl("srgan_comm_unique_id")) return e;
  return check_nccl(rccl().GetUniqueId(static_cast<ncclUniqueId*>(id128)), "ncclGetUniqueId");
}

extern "C" int srgan_comm_init(const void* id128, int nranks, int rank, void** comm) {
  SRGAN_REQUIRE(id128 && comm, "srgan_comm_init: null argument");
  SRGAN_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "srgan_comm_init: rank %d of %d", rank, nranks);
  if (int e = need_rccl("srgan_comm_init")) return e;
  ncclComm_t c = nullptr;
  ncclUniqueId id = *static_cast<const ncclUniqueId*>(id128);
  if (int e = check_nccl(rccl().CommInitRank(&c, nranks, id, rank), "ncclCommInitRank")) return e;
  *comm = c;
  return 0;
}

extern "C" int srgan_comm_destroy(void* comm) {
  if (!comm) return 0;
  if (int e = need_rccl("srgan_comm_destroy")) return e;
  return check_nccl(rccl().CommDestroy(static_cast<ncclComm_t>(comm)), "ncclCommDestroy");
}

extern "C" int srgan_comm_size(void* comm, int* nranks) {
  SRGAN_REQUIRE(comm && nranks, "srgan_comm_size: null argument");
  if (int e = need_rccl("srgan_comm_size")) return e;
  return check_nccl(rccl().CommCount(static_cast<ncclComm_t>(comm), nranks), "ncclCommCount");
}

extern "C" int srgan_allreduce_bucket(void* comm, void* buf, long long count, int bf16, int average, void* stream) {
  SRGAN_REQUIRE(comm, "srgan_allreduce_bucket: no communicator");
  SRGAN_REQUIRE(buf && count > 0, "srgan_allreduce_bucket: empty bucket");
  SRGAN_REQUIRE(bf16 == 0 || bf16 == 1, "srgan_allreduce_bucket: dtype flag %d", bf16);
  if (int e = need_rccl("srgan_allreduce_bucket")) return e;
  return check_nccl(rccl().AllReduce(buf, buf, (size_t)count, bf16 ? kNcclBfloat16 : kNcclFloat32, average ? kNcclAvg : kNcclSum,
                                     static_cast<ncclComm_t>(comm), as_stream(stream)),
                    "ncclAllReduce");
}

extern "C" int srgan_allgather_rows(void* comm, const float* rows, float* all_rows, long long count_per_rank, void* stream) {
  SRGAN_REQUIRE(comm, "srgan_allgather_rows: no communicator");
  SRGAN_REQUIRE(rows && all_rows && count_per_rank > 0, "srgan_allgather_rows: empty message");
  if (int e = need_rccl("srgan_allgather_rows")) return e;
  return check_nccl(rccl().AllGather(rows, all_rows, (size_t)count_per_rank, kNcclFloat32, static_cast<ncclComm_t>(comm),
                                     as_stream(stream)),
                    "ncclAllGather");
}
