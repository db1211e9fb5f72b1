// K13 behind the C ABI (SURVEY.md §8b "icv_allgather_kv"): the per-layer exchange of the post-RoPE K|V rows among the
// ranks of a sequence-parallel group as ONE RCCL all-gather per row chunk, for a host that does not go through
// torch.distributed.  RCCL is resolved at first use with dlopen (the process usually has PyTorch's librccl mapped already),
// so libicvideo.so itself carries no link-time dependency on it and loads on a box without RCCL.
//   rank 0 of the group:  icv_comm_unique_id(id)  -> the host ships the 128 bytes to the other ranks (any side channel)
//   every rank:           icv_comm_create(id, rank, world, &comm)   (ncclCommInitRank on the CURRENT device)
//   per chunk:            icv_allgather_kv(comm, rows, out, m, row_bytes, stream)   rows [m, row_bytes] -> out [world*m, row_bytes]
// The transfer is enqueued on the stream the caller passes: hand it a side stream and fence with events to overlap the
// exchange with attention (infinicube_amd/videogen/seqpar.py KVGather mode "native" does exactly that).
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "icv_common.h"

namespace {

typedef void* nccl_comm_t;
struct nccl_uid { char internal[ICV_COMM_ID_BYTES]; };
typedef int (*fn_get_uid)(nccl_uid*);
typedef int (*fn_init_rank)(nccl_comm_t*, int, nccl_uid, int);
typedef int (*fn_destroy)(nccl_comm_t);
typedef int (*fn_allgather)(const void*, void*, size_t, int /*ncclInt8 = 0*/, nccl_comm_t, hipStream_t);
typedef const char* (*fn_errstr)(int);

struct Rccl {
  void* lib = nullptr;
  fn_get_uid get_uid = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_destroy destroy = nullptr;
  fn_allgather allgather = nullptr;
  fn_errstr errstr = nullptr;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* nme : names) {
      r.lib = dlopen(nme, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) return;
    r.get_uid = (fn_get_uid)dlsym(r.lib, "ncclGetUniqueId");
    r.init_rank = (fn_init_rank)dlsym(r.lib, "ncclCommInitRank");
    r.destroy = (fn_destroy)dlsym(r.lib, "ncclCommDestroy");
    r.allgather = (fn_allgather)dlsym(r.lib, "ncclAllGather");
    r.errstr = (fn_errstr)dlsym(r.lib, "ncclGetErrorString");
    if (!r.get_uid || !r.init_rank || !r.destroy || !r.allgather) {
      dlclose(r.lib);
      r.lib = nullptr;
    }
  });
  return r.lib ? &r : nullptr;
}

int fail(Rccl* r, const char* what, int rc) {
  icv_set_error("%s: RCCL error %d (%s)", what, rc, r->errstr ? r->errstr(rc) : "?");
  return 2;
}

}  // namespace

struct icv_comm {
  nccl_comm_t comm = nullptr;
  int rank = 0, world = 1;
};

extern "C" int icv_comm_unique_id(char* id) {
  ICV_REQUIRE(id, "icv_comm_unique_id: null argument");
  Rccl* r = rccl();
  ICV_REQUIRE(r, "icv_comm_unique_id: librccl.so not found (dlopen)");
  nccl_uid u;
  const int rc = r->get_uid(&u);
  if (rc) return fail(r, "ncclGetUniqueId", rc);
  memcpy(id, u.internal, ICV_COMM_ID_BYTES);
  return 0;
}

extern "C" int icv_comm_create(const char* id, int rank, int world, icv_comm** out) {
  ICV_REQUIRE(id && out, "icv_comm_create: null argument");
  ICV_REQUIRE(world >= 1 && rank >= 0 && rank < world, "icv_comm_create: bad (rank, world) = (%d, %d)", rank, world);
  Rccl* r = rccl();
  ICV_REQUIRE(r, "icv_comm_create: librccl.so not found (dlopen)");
  nccl_uid u;
  memcpy(u.internal, id, ICV_COMM_ID_BYTES);
  nccl_comm_t c = nullptr;
  const int rc = r->init_rank(&c, world, u, rank);
  if (rc) return fail(r, "ncclCommInitRank", rc);
  icv_comm* h = new icv_comm();
  h->comm = c; h->rank = rank; h->world = world;
  *out = h;
  return 0;
}

extern "C" void icv_comm_destroy(icv_comm* c) {
  if (!c) return;
  Rccl* r = rccl();
  if (r && c->comm) (void)r->destroy(c->comm);
  delete c;
}

extern "C" int icv_allgather_kv(icv_comm* c, const void* rows, void* out, int64_t m, int64_t row_bytes, void* stream) {
  ICV_REQUIRE(c && rows && out, "icv_allgather_kv: null argument");
  ICV_REQUIRE(m > 0 && row_bytes > 0, "icv_allgather_kv: empty chunk");
  Rccl* r = rccl();
  ICV_REQUIRE(r, "icv_allgather_kv: librccl.so not found (dlopen)");
  const int rc = r->allgather(rows, out, (size_t)(m * row_bytes), 0, c->comm, (hipStream_t)stream);
  if (rc) return fail(r, "ncclAllGather", rc);
  return 0;
}
