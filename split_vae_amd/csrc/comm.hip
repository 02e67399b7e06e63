// sv_comm_*: the data-parallel gradient exchange of the SPLIT-VAE step as a C ABI over RCCL.
//
// The reference has no distributed code; the step shards over the batch axis and needs ONE collective:
// an in-place all-reduce(sum) of (buckets of) the flat fp32 gradient buffer, enqueued on a HIP stream so that it
// overlaps the rest of the backward pass; the 1/world factor lives in sv_adam_step (grad_scale).
//
// RCCL is bound at run time (dlopen), not at link time: a process that already carries an RCCL -- PyTorch ships its own
// librccl.so -- must not get a second copy with separate global state, and single-GPU users need no RCCL at all.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/splitvae.h"

namespace {

typedef struct { char internal[128]; } rccl_unique_id;      // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rccl_comm_t;
enum { RCCL_SUM = 0, RCCL_FLOAT32 = 7 };                    // ncclSum, ncclFloat32

struct Api {
  int (*GetUniqueId)(rccl_unique_id*) = nullptr;
  int (*CommInitRank)(rccl_comm_t*, int, rccl_unique_id, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
  int (*CommDestroy)(rccl_comm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  bool ok = false;
};

const Api& api() {
  static const Api a = [] {
    Api r;
    void* h = nullptr;
    // an RCCL that is already in the process first (RTLD_NOLOAD), then the system one
    const char* names[] = {"librccl.so", "librccl.so.1", nullptr};
    for (int i = 0; names[i] && !h; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);
    const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", nullptr};
    for (int i = 0; paths[i] && !h; ++i) h = dlopen(paths[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) return r;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
    r.ok = r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy && r.GroupStart && r.GroupEnd;
    return r;
  }();
  return a;
}

}  // namespace

struct sv_comm {
  rccl_comm_t comm;
  int rank, world;
};

extern "C" int sv_comm_unique_id(void* id128) {
  if (!id128) return SV_E_BADARG;
  if (!api().ok) return SV_E_UNSUPPORTED;
  rccl_unique_id id;
  const int rc = api().GetUniqueId(&id);
  if (rc) return SV_E_STATE;
  memcpy(id128, &id, sizeof(id));
  return SV_OK;
}

extern "C" int sv_comm_init(const void* id128, int32_t rank, int32_t world, sv_comm** out) {
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return SV_E_BADARG;
  if (!api().ok) return SV_E_UNSUPPORTED;
  rccl_unique_id id;
  memcpy(&id, id128, sizeof(id));
  rccl_comm_t c = nullptr;
  if (api().CommInitRank(&c, world, id, rank) != 0 || !c) return SV_E_STATE;   // binds the calling thread's current device
  *out = new sv_comm{c, rank, world};
  return SV_OK;
}

extern "C" int sv_comm_allreduce(sv_comm* c, float* buf, int64_t count, void* stream) {
  if (!c || !buf || count < 0) return SV_E_BADARG;
  if (count == 0) return SV_OK;
  return api().AllReduce(buf, buf, (size_t)count, RCCL_FLOAT32, RCCL_SUM, c->comm, (hipStream_t)stream) == 0 ? SV_OK : SV_E_STATE;
}

// several disjoint ranges of one buffer as ONE RCCL group (one launch on the stream)
extern "C" int sv_comm_allreduce_ranges(sv_comm* c, float* base, const int64_t* begin, const int64_t* end, int32_t n, void* stream) {
  if (!c || !base || (n > 0 && (!begin || !end)) || n < 0) return SV_E_BADARG;
  int bad = 0;
  if (api().GroupStart() != 0) return SV_E_STATE;
  for (int i = 0; i < n; ++i) {
    if (end[i] < begin[i]) { bad = 1; continue; }
    if (end[i] == begin[i]) continue;
    bad |= api().AllReduce(base + begin[i], base + begin[i], (size_t)(end[i] - begin[i]), RCCL_FLOAT32, RCCL_SUM, c->comm,
                           (hipStream_t)stream) != 0;
  }
  if (api().GroupEnd() != 0 || bad) return SV_E_STATE;
  return SV_OK;
}

extern "C" int sv_comm_destroy(sv_comm* c) {
  if (!c) return SV_OK;
  const int rc = api().ok ? api().CommDestroy(c->comm) : 0;
  delete c;
  return rc == 0 ? SV_OK : SV_E_STATE;
}
