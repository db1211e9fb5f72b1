// K13 with the rows moved by the copy engines (SURVEY.md §8e "hide it"): the per-layer exchange of the post-RoPE K|V rows as PULLS.
// An RCCL all-gather is a kernel: every channel is a work-group pinned on a CU for the whole transfer, and the attention it is
// supposed to hide under runs exactly one 8-wave work-group per CU — a taken CU stretches a whole round of that launch
// (measured: profiles/r05/kv_contention.md).  Here no kernel moves a row; what does run on a CU is one wave per flag operation
// (see "MEASURED" below), resident for the skew between two ranks instead of for the transfer:
//
//   * every rank keeps its K|V rows in a SYMMETRIC HEAP — one device buffer of the same size on every rank (the caller's own,
//     e.g. a torch tensor: the allocation that contains it is found with hipMemGetAddressRange and exported whole, the offset
//     travels with the handle — what torch's own CUDA-IPC does; or a hipMalloc made here), exported once with
//     hipIpcGetMemHandle and opened by every peer (hipIpcOpenMemHandle, peer access enabled lazily), so "rows at byte offset
//     o of rank p" is an address this rank can hand to hipMemcpyAsync; between two devices that copy is executed by an SDMA
//     engine over the xGMI link of that pair, and the (world - 1) pulls of a chunk run on (world - 1) streams — one per peer =
//     one per link (xGMI is point-to-point: 7 links, 7 engines, no ring);
//   * readiness is a FLAG WORD per (rank, slot) in a small POSIX shared-memory segment that every rank maps and registers
//     with hipHostRegister: the producer's launch stream writes a sequence number behind the kernels that produced the rows
//     (hipStreamWriteValue32), the consumer's pull stream waits for ">= that number" in front of its copy — no host round trip.
//     MEASURED on this runtime (tools/probe_streamops.py under rocprofv3): the runtime's write is a 1-2 us kernel
//     (__amd_rocclr_streamOpsWrite) and its wait (hipStreamWaitValue32) a ONE-WAVE KERNEL THAT SPINS (__amd_rocclr_streamOpsWait), not a
//     command-processor packet - so since round 6 the wait is a one-wave kernel of OURS with a deadline (wait_ready_kernel below): the
//     same cost, resident only for the skew between the two ranks (they run the same program), but a dead peer is a time-out + an error
//     word instead of a queue that spins for ever.  It is the rows, not the flags, whose movement would otherwise occupy CUs for the whole
//     transfer.  Sequence numbers only grow, so a late waiter can never miss a publish and no host handshake is needed;
//   * per peer a DEVICE word arrived[p] = ticket + 1 is stored behind each landed copy: the arrival-driven attention
//     (icv_attention_fwd_pieces, csrc/attn7p.hip) polls it from inside ONE launch per layer instead of the host launching per chunk;
//   * a rank's OWN rows reach its gathered buffer by a local copy on a stream of its own (behind the same point of the launch
//     stream as the pulls), so nothing of the exchange sits in the launch stream between two projections;
//   * the reverse hazard (the producer's next layer overwriting rows a slow peer is still pulling) is closed the same way:
//     each pull stream writes "pulled up to ticket k" into done[consumer][producer] behind its copy, and
//     icv_ipc_acquire makes the producer's launch stream wait for every peer's counter before the K|V GEMM of the next layer.
//
// Every rank issues the same sequence of icv_ipc_gather_start calls (same program), so ticket k names the same exchange
// everywhere and its flag slot / sequence number need no negotiation.  Deadlock freedom with streams multiplexed onto a few
// hardware queues: every wait a rank enqueues is for a flag whose write the peer enqueued EARLIER in its own program order
// than its own waits of the same ticket, so processing each rank's packets in submission order always terminates.
//
// The host side (infinicube_amd/videogen/seqpar.py, KVGather mode "ipc") ships the 72-byte handles and the segment name
// through the torch.distributed group once, runs a pattern self-test, and joins the start-up autotune as one more candidate.
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <stdlib.h>
#include <unistd.h>

#include <vector>

#include "icv_common.h"

namespace {

constexpr int kSlots = ICV_IPC_SLOTS;

#define ICV_HIP_OK(call, what)                                                          \
  do {                                                                                  \
    const hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) {                                                             \
      icv_set_error("%s: %s (%s)", what, hipGetErrorName(e_), hipGetErrorString(e_));   \
      return 2;                                                                         \
    }                                                                                   \
  } while (0)

// segment layout (uint32 words): ready[world][kSlots] | done[world][world] | err[world]
size_t flags_bytes(int world) { return sizeof(uint32_t) * ((size_t)world * kSlots + (size_t)world * world + (size_t)world); }

// Every wait of this transport is a kernel of OURS with a deadline (round 6; rounds 5's pull streams used hipStreamWaitValue32, which on
// this runtime is a one-wave kernel that spins for ever - measured, tools/probe_streamops.py - so a peer that died mid-run left the
// survivors' queues spinning and their teardown blocked).  Same cost (one wave, resident for the skew between two ranks), but after
// `timeout_ticks` (100 MHz s_memrealtime) it records WHO it was waiting for in this rank's error word and returns: the copy behind it
// then moves stale rows, the host's next icv_ipc_check raises, and nothing hangs.
constexpr uint32_t kErrReady = 1u << 30, kErrDone = 2u << 30;     // err word: kind | peer << 20 | (sequence number & 0xfffff)

// arrival flag in DEVICE memory (what icv_attention_fwd_pieces polls): a system-scope release store behind the copy on its stream
__global__ void publish_kernel(uint32_t* flag, uint32_t value) { __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

__global__ void wait_ready_kernel(const uint32_t* flag, uint32_t value, unsigned long long timeout_ticks, uint32_t* err, uint32_t code) {
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) {
    __builtin_amdgcn_s_sleep(8);
    // fail fast: after the FIRST wait of this rank that gave up, later waits do not sit out the deadline again (a dead peer would
    // otherwise cost deadline x chunks x layers before the host's next icv_ipc_check)
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return;
    if (timeout_ticks && __builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
      atomicCAS_system(err, 0u, code);
      return;
    }
  }
}

// icv_ipc_acquire: ONE launch in which lane p waits for peer p's "pulled up to ticket k" counter (the runtime's hipStreamWaitValue32 is
// a spinning kernel per call as well - measured - so (world - 1) of them would be (world - 1) launches per layer on the launch stream)
__global__ void wait_done_kernel(const uint32_t* done_col, int world, int rank, uint32_t value, unsigned long long timeout_ticks, uint32_t* err) {
  const int p = threadIdx.x;
  if (p >= world || p == rank) return;
  const uint32_t* f = done_col + (size_t)p * world;          // done[p][rank]
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((int32_t)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) {
    __builtin_amdgcn_s_sleep(8);
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return;       // fail fast (see wait_ready_kernel)
    if (timeout_ticks && __builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
      atomicCAS_system(err, 0u, kErrDone | ((uint32_t)p << 20) | (value & 0xfffffu));
      return;
    }
  }
}

}  // namespace

struct icv_ipc {
  int rank = 0, world = 1, device = 0;
  char* heap = nullptr;
  int64_t heap_bytes = 0;
  bool own_heap = false;               // hipMalloc'ed here (heap == NULL at create) or borrowed from the caller
  std::vector<char*> peer;             // address of every rank's heap as THIS process sees it ([rank] = own heap)
  std::vector<void*> peer_base;        // what hipIpcOpenMemHandle returned (the allocation the peer's heap lives in)
  std::vector<hipStream_t> pull;       // one pull stream per peer; [rank] = the stream of this rank's own rows (a local copy)
  std::vector<hipEvent_t> landed;      // [peer * kSlots + slot]: that peer's chunk of the ticket in this slot has landed
  hipEvent_t started[kSlots] = {};     // launch-stream position at gather_start (the pulls' destination is free from there)
  uint32_t* flags_host = nullptr;      // mmap of the shared segment
  uint32_t* flags_dev = nullptr;       // the same words as the GPU addresses them
  size_t flags_len = 0;
  bool registered = false;
  int64_t next_ticket = 0;
  bool waited[kSlots];
  bool aborted = false;
  hipStream_t last_stream = nullptr;   // the launch stream of the latest gather_start (icv_ipc_abort drains it)
  bool last_stream_valid = false;
  unsigned long long timeout_ticks = 0;   // deadline of every device-side wait (100 MHz ticks; 0 = none)
  int drain_ms = 5000;                    // how long teardown lets this rank's queues finish before it releases every wait
  bool copy_own_rows = true;              // gather_start also copies this rank's own rows into `out` (off for the arrival-driven attention, which reads them in place)
  uint32_t* arrived = nullptr;            // DEVICE memory [world]: arrived[p] = t + 1 once peer p's rows of ticket t have landed (monotonic)
  uint32_t* err_word() const { return flags_dev + (size_t)world * kSlots + (size_t)world * world + rank; }
  volatile uint32_t* err_host(int r) const { return flags_host + (size_t)world * kSlots + (size_t)world * world + r; }
  uint32_t* ready(int r, int slot) const { return flags_dev + (size_t)r * kSlots + slot; }
  uint32_t* done(int consumer, int producer) const { return flags_dev + (size_t)world * kSlots + (size_t)consumer * world + producer; }
};

extern "C" int icv_ipc_create(const char* shm_name, int rank, int world, void* heap_ptr, int64_t heap_bytes, icv_ipc** out) {
  ICV_REQUIRE(shm_name && out, "icv_ipc_create: null argument");
  ICV_REQUIRE(world >= 1 && rank >= 0 && rank < world, "icv_ipc_create: bad (rank, world) = (%d, %d)", rank, world);
  ICV_REQUIRE(heap_bytes > 0, "icv_ipc_create: empty heap");
  int dev = 0, can_wait = 0;
  ICV_HIP_OK(hipGetDevice(&dev), "hipGetDevice");
  ICV_HIP_OK(hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, dev), "hipDeviceGetAttribute");
  ICV_REQUIRE(can_wait, "icv_ipc_create: device %d cannot execute hipStreamWaitValue32 (no copy-engine K|V transport here)", dev);
  icv_ipc* c = new icv_ipc();
  c->rank = rank; c->world = world; c->device = dev; c->heap_bytes = heap_bytes;
  {   // deadlines (milliseconds): ICV_IPC_WAIT_TIMEOUT_MS bounds every device-side wait (0 = wait for ever), ICV_IPC_DRAIN_TIMEOUT_MS the teardown
    const char* w = getenv("ICV_IPC_WAIT_TIMEOUT_MS");
    const char* d = getenv("ICV_IPC_DRAIN_TIMEOUT_MS");
    const long long wait_ms = w ? atoll(w) : 60000;
    c->timeout_ticks = wait_ms > 0 ? (unsigned long long)wait_ms * 100000ull : 0ull;
    if (d) c->drain_ms = atoi(d) > 0 ? atoi(d) : 0;
  }
  c->peer.assign(world, nullptr);
  for (int s = 0; s < kSlots; ++s) c->waited[s] = true;
  auto fail = [&](int rc) { icv_ipc_destroy(c); return rc; };
  // flag segment: created by whoever comes first (new POSIX shared memory reads as zeros), sized identically by everyone
  const int fd = shm_open(shm_name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) { icv_set_error("icv_ipc_create: shm_open(%s): %s", shm_name, strerror(errno)); return fail(2); }
  c->flags_len = flags_bytes(world);
  if (ftruncate(fd, (off_t)c->flags_len) != 0) { icv_set_error("icv_ipc_create: ftruncate: %s", strerror(errno)); close(fd); return fail(2); }
  void* m = mmap(nullptr, c->flags_len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) { icv_set_error("icv_ipc_create: mmap: %s", strerror(errno)); return fail(2); }
  c->flags_host = (uint32_t*)m;
#define ICV_IPC_TRY(call, what)                                                              \
  do {                                                                                       \
    const hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                                  \
      icv_set_error("icv_ipc_create: %s: %s (%s)", what, hipGetErrorName(e_), hipGetErrorString(e_)); \
      return fail(2);                                                                        \
    }                                                                                        \
  } while (0)
  ICV_IPC_TRY(hipHostRegister(m, c->flags_len, hipHostRegisterMapped | hipHostRegisterPortable), "hipHostRegister(flag segment)");
  c->registered = true;
  void* d = nullptr;
  ICV_IPC_TRY(hipHostGetDevicePointer(&d, m, 0), "hipHostGetDevicePointer");
  c->flags_dev = (uint32_t*)d;
  void* heap = heap_ptr;
  if (!heap) {
    ICV_IPC_TRY(hipMalloc(&heap, (size_t)heap_bytes), "hipMalloc(symmetric heap)");
    c->own_heap = true;
  }
  c->heap = (char*)heap;
  c->peer[rank] = c->heap;
  c->pull.assign(world, nullptr);
  c->landed.assign((size_t)world * kSlots, nullptr);
  for (int p = 0; p < world; ++p) {      // p == rank too: this rank's own rows travel on their own stream, off the launch stream
    ICV_IPC_TRY(hipStreamCreateWithFlags(&c->pull[p], hipStreamNonBlocking), "hipStreamCreateWithFlags");
    for (int s = 0; s < kSlots; ++s) ICV_IPC_TRY(hipEventCreateWithFlags(&c->landed[(size_t)p * kSlots + s], hipEventDisableTiming), "hipEventCreateWithFlags");
  }
  for (int s = 0; s < kSlots; ++s) ICV_IPC_TRY(hipEventCreateWithFlags(&c->started[s], hipEventDisableTiming), "hipEventCreateWithFlags");
  ICV_IPC_TRY(hipMalloc((void**)&c->arrived, sizeof(uint32_t) * (size_t)world), "hipMalloc(arrival flags)");
  ICV_IPC_TRY(hipMemset(c->arrived, 0, sizeof(uint32_t) * (size_t)world), "hipMemset(arrival flags)");
#undef ICV_IPC_TRY
  *out = c;
  return 0;
}

extern "C" int icv_ipc_shm_unlink(const char* shm_name) {
  ICV_REQUIRE(shm_name, "icv_ipc_shm_unlink: null argument");
  if (shm_unlink(shm_name) != 0 && errno != ENOENT) {
    icv_set_error("icv_ipc_shm_unlink(%s): %s", shm_name, strerror(errno));
    return 2;
  }
  return 0;
}

extern "C" int icv_ipc_heap(icv_ipc* c, void** base, int64_t* bytes) {
  ICV_REQUIRE(c && base && bytes, "icv_ipc_heap: null argument");
  *base = c->heap;
  *bytes = c->heap_bytes;
  return 0;
}

extern "C" int icv_ipc_export(icv_ipc* c, char* handle) {
  ICV_REQUIRE(c && handle, "icv_ipc_export: null argument");
  static_assert(sizeof(hipIpcMemHandle_t) + sizeof(int64_t) == ICV_IPC_HANDLE_BYTES, "ICV_IPC_HANDLE_BYTES = hipIpcMemHandle_t + offset");
  // the handle names a whole ALLOCATION: find the one that contains the heap and ship the heap's offset inside it
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  ICV_HIP_OK(hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)c->heap), "hipMemGetAddressRange(symmetric heap)");
  const int64_t off = c->heap - (char*)base;
  ICV_REQUIRE(off >= 0 && (size_t)(off + c->heap_bytes) <= size, "icv_ipc_export: the heap [%p, +%lld) is not inside one device allocation (base %p, %zu bytes)",
              (void*)c->heap, (long long)c->heap_bytes, (void*)base, size);
  hipIpcMemHandle_t h;
  ICV_HIP_OK(hipIpcGetMemHandle(&h, (void*)base), "hipIpcGetMemHandle");
  memcpy(handle, &h, sizeof(h));
  memcpy(handle + sizeof(h), &off, sizeof(off));
  return 0;
}

extern "C" int icv_ipc_open_peer(icv_ipc* c, int peer, const char* handle) {
  ICV_REQUIRE(c && handle, "icv_ipc_open_peer: null argument");
  ICV_REQUIRE(peer >= 0 && peer < c->world && peer != c->rank, "icv_ipc_open_peer: bad peer %d (rank %d of %d)", peer, c->rank, c->world);
  ICV_REQUIRE(!c->peer[peer], "icv_ipc_open_peer: peer %d is already open", peer);
  hipIpcMemHandle_t h;
  int64_t off = 0;
  memcpy(&h, handle, sizeof(h));
  memcpy(&off, handle + sizeof(h), sizeof(off));
  void* p = nullptr;
  ICV_HIP_OK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle");
  c->peer_base.resize(c->world, nullptr);
  c->peer_base[peer] = p;
  c->peer[peer] = (char*)p + off;
  return 0;
}

extern "C" int icv_ipc_abort(icv_ipc* c);

extern "C" int icv_ipc_gather_start(icv_ipc* c, int64_t src_offset, int64_t bytes, void* out, void* stream, int64_t* ticket) {
  ICV_REQUIRE(c && out && ticket, "icv_ipc_gather_start: null argument");
  ICV_REQUIRE(bytes > 0 && src_offset >= 0 && src_offset + bytes <= c->heap_bytes,
              "icv_ipc_gather_start: rows [%lld, +%lld) are outside the %lld-byte symmetric heap", (long long)src_offset, (long long)bytes,
              (long long)c->heap_bytes);
  for (int p = 0; p < c->world; ++p) ICV_REQUIRE(c->peer[p], "icv_ipc_gather_start: peer %d's heap was never opened", p);
  ICV_REQUIRE(!c->aborted, "icv_ipc_gather_start: this transport was aborted (icv_ipc_abort)");
  const int64_t k = c->next_ticket;
  const int slot = (int)(k % kSlots);
  ICV_REQUIRE(c->waited[slot], "icv_ipc_gather_start: %d exchanges in flight without a wait (ticket %lld was never waited for)", kSlots,
              (long long)(k - kSlots));
  const uint32_t seq = (uint32_t)(k / kSlots + 1);
  hipStream_t s = (hipStream_t)stream;
  char* dst = (char*)out;
  c->last_stream = s;
  c->last_stream_valid = true;
  // a runtime failure from here on leaves this rank's exchange half-enqueued while the peers wait for it: release them (icv_ipc_abort)
#define ICV_IPC_RUN(call, what)                                                                                  \
  do {                                                                                                           \
    const hipError_t e_ = (call);                                                                                \
    if (e_ != hipSuccess) {                                                                                      \
      (void)icv_ipc_abort(c);                                                                                    \
      icv_set_error("icv_ipc_gather_start: %s: %s (%s); transport aborted", what, hipGetErrorName(e_), hipGetErrorString(e_)); \
      return 2;                                                                                                  \
    }                                                                                                            \
  } while (0)
  // the rows were produced on `s`: publish them behind their producers
  ICV_IPC_RUN(hipStreamWriteValue32(s, c->ready(c->rank, slot), seq, 0), "hipStreamWriteValue32(ready)");
  // everything this rank still does with `out` (the previous layer's attention reads it) was enqueued on `s` before this point
  ICV_IPC_RUN(hipEventRecord(c->started[slot], s), "hipEventRecord(started)");
  for (int i = 1; i < c->world; ++i) {
    const int p = (c->rank + i) % c->world;          // start with the right-hand neighbour: at any moment every link is asked once
    hipStream_t ps = c->pull[p];
    ICV_IPC_RUN(hipStreamWaitEvent(ps, c->started[slot], 0), "hipStreamWaitEvent(started)");
    hipLaunchKernelGGL(wait_ready_kernel, dim3(1), dim3(64), 0, ps, (const uint32_t*)c->ready(p, slot), seq, c->timeout_ticks, c->err_word(),
                       kErrReady | ((uint32_t)p << 20) | (seq & 0xfffffu));
    ICV_IPC_RUN(hipGetLastError(), "wait_ready_kernel");
    ICV_IPC_RUN(hipMemcpyAsync(dst + (int64_t)p * bytes, c->peer[p] + src_offset, (size_t)bytes, hipMemcpyDeviceToDevice, ps), "hipMemcpyAsync(pull)");
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(1), 0, ps, c->arrived + p, (uint32_t)(k + 1));     // device word: the attention polls it
    ICV_IPC_RUN(hipGetLastError(), "publish_kernel(arrived)");
    ICV_IPC_RUN(hipStreamWriteValue32(ps, c->done(c->rank, p), (uint32_t)(k + 1), 0), "hipStreamWriteValue32(done)");
    ICV_IPC_RUN(hipEventRecord(c->landed[(size_t)p * kSlots + slot], ps), "hipEventRecord(landed)");
  }
  // own rows: a local copy (a short blit kernel) behind the same `started` point, on its own stream - on the launch stream it sat
  // between the K|V projection and the Q projection of every layer (2 % of a layer at the sp8 shard shapes)
  {
    hipStream_t ps = c->pull[c->rank];
    ICV_IPC_RUN(hipStreamWaitEvent(ps, c->started[slot], 0), "hipStreamWaitEvent(started)");
    if (c->copy_own_rows)
      ICV_IPC_RUN(hipMemcpyAsync(dst + (int64_t)c->rank * bytes, c->heap + src_offset, (size_t)bytes, hipMemcpyDeviceToDevice, ps), "hipMemcpyAsync(own rows)");
    ICV_IPC_RUN(hipEventRecord(c->landed[(size_t)c->rank * kSlots + slot], ps), "hipEventRecord(landed)");
  }
  c->waited[slot] = false;
  c->next_ticket = k + 1;
  *ticket = k;
  return 0;
#undef ICV_IPC_RUN
}

extern "C" int icv_ipc_gather_wait(icv_ipc* c, int64_t ticket, void* stream) {
  ICV_REQUIRE(c, "icv_ipc_gather_wait: null argument");
  ICV_REQUIRE(ticket >= 0 && ticket < c->next_ticket && ticket >= c->next_ticket - kSlots, "icv_ipc_gather_wait: ticket %lld is not in flight (next %lld)",
              (long long)ticket, (long long)c->next_ticket);
  const int slot = (int)(ticket % kSlots);
  for (int p = 0; p < c->world; ++p)      // every peer's chunk and this rank's own
    ICV_HIP_OK(hipStreamWaitEvent((hipStream_t)stream, c->landed[(size_t)p * kSlots + slot], 0), "hipStreamWaitEvent(landed)");
  c->waited[slot] = true;
  return 0;
}

// The consumer gated on the arrival flags itself (icv_ipc_arrival + icv_attention_fwd_pieces): bookkeeping only, nothing is enqueued.
// (Not "icv_ipc_gather_wait with a NULL stream": NULL IS a stream - the default one.)
extern "C" int icv_ipc_gather_consumed(icv_ipc* c, int64_t ticket) {
  ICV_REQUIRE(c, "icv_ipc_gather_consumed: null argument");
  ICV_REQUIRE(ticket >= 0 && ticket < c->next_ticket && ticket >= c->next_ticket - kSlots, "icv_ipc_gather_consumed: ticket %lld is not in flight (next %lld)",
              (long long)ticket, (long long)c->next_ticket);
  c->waited[(int)(ticket % kSlots)] = true;
  return 0;
}

extern "C" int icv_ipc_acquire(icv_ipc* c, void* stream) {
  ICV_REQUIRE(c, "icv_ipc_acquire: null argument");
  if (c->next_ticket == 0) return 0;
  // this rank's own copies read the heap rows too (their stream is in order: the latest one covers the earlier ones)
  ICV_HIP_OK(hipStreamWaitEvent((hipStream_t)stream, c->landed[(size_t)c->rank * kSlots + (int)((c->next_ticket - 1) % kSlots)], 0),
             "hipStreamWaitEvent(own rows read)");
  if (c->world == 1) return 0;
  ICV_REQUIRE(c->world <= 64, "icv_ipc_acquire: at most 64 ranks");
  hipLaunchKernelGGL(wait_done_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint32_t*)c->done(0, c->rank), c->world, c->rank, (uint32_t)c->next_ticket,
                     c->timeout_ticks, c->err_word());
  return icv_check_launch("icv_ipc_acquire");
}

extern "C" int icv_ipc_arrival(icv_ipc* c, const uint32_t** flags) {
  ICV_REQUIRE(c && flags, "icv_ipc_arrival: null argument");
  *flags = c->arrived;
  return 0;
}

extern "C" int icv_ipc_configure(icv_ipc* c, int copy_own_rows) {
  ICV_REQUIRE(c, "icv_ipc_configure: null argument");
  c->copy_own_rows = copy_own_rows != 0;
  return 0;
}

// Did one of this rank's device-side waits give up?  (The host calls this at its own pace - once per denoising step is enough: a wait that
// timed out has already let its queue go on.)  Non-zero + text = the transport moved stale rows: the run is invalid from that exchange on.
extern "C" int icv_ipc_check(icv_ipc* c) {
  ICV_REQUIRE(c, "icv_ipc_check: null argument");
  if (!c->flags_host) return 0;
  const uint32_t e = *c->err_host(c->rank);
  if (!e) return 0;
  const int peer = (int)((e >> 20) & 0x3ffu);
  icv_set_error("copy-engine K|V transport: rank %d gave up waiting for rank %d to %s (sequence %u) after %.1f s - that rank is dead or stalled; "
                "the rows used since then are stale", c->rank, peer, (e & kErrDone) == kErrDone ? "finish pulling this rank's rows" : "publish its rows",
                e & 0xfffffu, (double)c->timeout_ticks / 1e8);
  return 3;
}

// A rank that cannot go on (a failed call in the middle of an exchange) must not leave its peers' queues spinning on flags it
// will never write: every flag word this rank owns jumps to a value every present and future waiter accepts (the waits compare
// ">= sequence number").  Peers then pull whatever bytes are there - the caller's own error report (the collective self-test, the
// launch ladder) is what stops the run; this only guarantees that it is an ERROR everywhere and not a hang somewhere.
// Flag writes this rank enqueued earlier (legitimate, smaller sequence numbers) may still execute AFTER the host's poison and
// overwrite it, so: poison, let this rank's own queues drain for a bounded time (they depend only on tickets the peers have
// published, or are released by the peers' own aborts), poison again.
extern "C" int icv_ipc_abort(icv_ipc* c) {
  ICV_REQUIRE(c, "icv_ipc_abort: null argument");
  c->aborted = true;
  if (!c->flags_host) return 0;
  volatile uint32_t* f = c->flags_host;
  constexpr uint32_t kPoison = 0x7fffffffu;
  auto poison = [&]() {
    for (int s = 0; s < kSlots; ++s) f[(size_t)c->rank * kSlots + s] = kPoison;                                   // ready[rank][*]
    for (int p = 0; p < c->world; ++p) f[(size_t)c->world * kSlots + (size_t)c->rank * c->world + p] = kPoison;   // done[rank][*]
    __sync_synchronize();
  };
  poison();
  for (int spin = 0; spin < 2000; ++spin) {        // <= 2 s
    bool busy = c->last_stream_valid && hipStreamQuery(c->last_stream) == hipErrorNotReady;
    for (hipStream_t ps : c->pull) busy = busy || (ps && hipStreamQuery(ps) == hipErrorNotReady);
    if (!busy) break;
    usleep(1000);
  }
  poison();
  return 0;
}

extern "C" int64_t icv_ipc_tickets(const icv_ipc* c) { return c ? c->next_ticket : -1; }

// ---- first-contact probe: does a pull from `peer` need compute units? ---------------------------------------------------------
// The point of this transport is that the ROWS move without a kernel.  Whether hipMemcpyAsync between two devices is executed by an
// SDMA engine or by a blit kernel (__amd_rocclr_copyBuffer) is the runtime's choice and nothing in a timing table would tell (a
// same-device pull IS a blit: profiles/r05/stream_ops_probe.txt).  The probe is self-validating: an occupier kernel takes EVERY
// wave slot of the device for ~8 ms (grid = the occupancy API's resident work-groups x CUs, a census counter confirms they all
// started); while it holds them a control kernel (one thread) and the copy are enqueued on two other streams:
//   control finished early            -> 0 = inconclusive (the device was not full: the occupancy figure was wrong)
//   copy finished, control did not    -> 1 = copy engine (no wave was needed)
//   neither finished                  -> 2 = the copy is a kernel waiting for a wave slot (blit)
namespace {
__global__ __launch_bounds__(1024) void occupy_kernel(unsigned long long ticks, int* census) {
  if (threadIdx.x == 0) atomicAdd(census, 1);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
__global__ void touch_kernel(int* flag) { *flag = 1; }
}  // namespace

// the probe for any (src, dst): device, peer-device or pinned host pointers; dst == NULL = a scratch device buffer
extern "C" int icv_probe_copy_path(const void* src, void* dst_arg, int64_t bytes, int* kind, double* copy_ms) {
  ICV_REQUIRE(src && kind && bytes > 0, "icv_probe_copy_path: bad argument");
  *kind = 0;
  if (copy_ms) *copy_ms = -1.0;
  int device = 0;
  ICV_HIP_OK(hipGetDevice(&device), "hipGetDevice");
  hipDeviceProp_t prop;
  ICV_HIP_OK(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
  int per_cu = 0;
  ICV_HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, occupy_kernel, 1024, 0), "hipOccupancyMaxActiveBlocksPerMultiprocessor");
  ICV_REQUIRE(per_cu > 0, "icv_probe_copy_path: the occupier does not fit a CU");
  const int nwg = per_cu * prop.multiProcessorCount;
  void* dst = dst_arg;
  int* words = nullptr;                 // [0] census, [1] control flag
  hipStream_t s_occ = nullptr, s_ctl = nullptr, s_cpy = nullptr;
  hipEvent_t e_occ = nullptr, e_ctl = nullptr, e_c0 = nullptr, e_c1 = nullptr;
  int rc = 0;
#define PROBE_TRY(call, what)                                                                 \
  do {                                                                                        \
    const hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess && rc == 0) {                                                        \
      icv_set_error("icv_probe_copy_path: %s: %s (%s)", what, hipGetErrorName(e_), hipGetErrorString(e_)); \
      rc = 2;                                                                                 \
    }                                                                                         \
  } while (0)
  if (!dst) PROBE_TRY(hipMalloc(&dst, (size_t)bytes), "hipMalloc(scratch)");
  PROBE_TRY(hipMalloc((void**)&words, 2 * sizeof(int)), "hipMalloc(words)");
  if (rc == 0) PROBE_TRY(hipMemset(words, 0, 2 * sizeof(int)), "hipMemset");
  PROBE_TRY(hipStreamCreateWithFlags(&s_occ, hipStreamNonBlocking), "hipStreamCreate");
  PROBE_TRY(hipStreamCreateWithFlags(&s_ctl, hipStreamNonBlocking), "hipStreamCreate");
  PROBE_TRY(hipStreamCreateWithFlags(&s_cpy, hipStreamNonBlocking), "hipStreamCreate");
  PROBE_TRY(hipEventCreate(&e_occ), "hipEventCreate");
  PROBE_TRY(hipEventCreate(&e_ctl), "hipEventCreate");
  PROBE_TRY(hipEventCreate(&e_c0), "hipEventCreate");
  PROBE_TRY(hipEventCreate(&e_c1), "hipEventCreate");
  if (rc == 0) {
    PROBE_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
    // warm both paths once (first-use set-up of a peer mapping / a blit kernel must not be mistaken for "waiting for a wave")
    PROBE_TRY(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDefault, s_cpy), "hipMemcpyAsync(warm-up)");
    hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(1), 0, s_ctl, words + 1);
    PROBE_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
    PROBE_TRY(hipMemset(words, 0, 2 * sizeof(int)), "hipMemset");
    PROBE_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
  }
  if (rc == 0) {
    hipLaunchKernelGGL(occupy_kernel, dim3(nwg), dim3(1024), 0, s_occ, 800000ull /* 8 ms */, words);
    PROBE_TRY(hipEventRecord(e_occ, s_occ), "hipEventRecord");
    usleep(1500);                                   // every occupier work-group is resident by now
    hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(1), 0, s_ctl, words + 1);
    PROBE_TRY(hipEventRecord(e_ctl, s_ctl), "hipEventRecord");
    PROBE_TRY(hipEventRecord(e_c0, s_cpy), "hipEventRecord");
    PROBE_TRY(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDefault, s_cpy), "hipMemcpyAsync(probe)");
    PROBE_TRY(hipEventRecord(e_c1, s_cpy), "hipEventRecord");
    // poll while the occupier is still resident: what completes in that window needed no wave slot
    bool copy_done = false, ctl_done = false;
    int polls_while_full = 0;
    for (int i = 0; i < 80 && rc == 0; ++i) {
      usleep(100);
      if (hipEventQuery(e_occ) == hipSuccess) break;              // the window is over: what was not done by the LAST poll was waiting for a slot
      ++polls_while_full;
      ctl_done = ctl_done || hipEventQuery(e_ctl) == hipSuccess;
      copy_done = copy_done || hipEventQuery(e_c1) == hipSuccess;
      if (ctl_done || copy_done) break;
    }
    PROBE_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
    int host_words[2] = {0, 0};
    PROBE_TRY(hipMemcpy(host_words, words, sizeof(host_words), hipMemcpyDeviceToHost), "hipMemcpy");
    if (rc == 0) {
      // inconclusive: the control kernel got a slot, not every occupier work-group ran, or the window was too short to judge (< 2 ms)
      if (ctl_done || host_words[0] != nwg || (!copy_done && polls_while_full < 20)) *kind = 0;
      else *kind = copy_done ? 1 : 2;
      float ms = 0.f;
      if (copy_ms && hipEventElapsedTime(&ms, e_c0, e_c1) == hipSuccess) *copy_ms = (double)ms;
    }
  }
#undef PROBE_TRY
  if (e_occ) (void)hipEventDestroy(e_occ);
  if (e_ctl) (void)hipEventDestroy(e_ctl);
  if (e_c0) (void)hipEventDestroy(e_c0);
  if (e_c1) (void)hipEventDestroy(e_c1);
  if (s_occ) (void)hipStreamDestroy(s_occ);
  if (s_ctl) (void)hipStreamDestroy(s_ctl);
  if (s_cpy) (void)hipStreamDestroy(s_cpy);
  if (dst && !dst_arg) (void)hipFree(dst);
  if (words) (void)hipFree(words);
  return rc;
}

extern "C" int icv_ipc_probe_copy(icv_ipc* c, int peer, int64_t bytes, int* kind, double* copy_ms) {
  ICV_REQUIRE(c && kind, "icv_ipc_probe_copy: null argument");
  ICV_REQUIRE(peer >= 0 && peer < c->world && c->peer[peer], "icv_ipc_probe_copy: peer %d is not open", peer);
  ICV_REQUIRE(bytes > 0 && bytes <= c->heap_bytes, "icv_ipc_probe_copy: %lld bytes do not fit the %lld-byte heap", (long long)bytes, (long long)c->heap_bytes);
  return icv_probe_copy_path(c->peer[peer], nullptr, bytes, kind, copy_ms);
}

// Teardown must not depend on the peers being alive: give this rank's queues `timeout_ms` to finish on their own, then satisfy EVERY wait
// word of the segment (this rank's queues only ever wait on words of the segment; a dead peer cannot object) so that whatever is still
// queued here runs to its end.  Returns 0 when the queues drained by themselves, 1 when the waits had to be released.
extern "C" int icv_ipc_drain(icv_ipc* c, int timeout_ms) {
  ICV_REQUIRE(c, "icv_ipc_drain: null argument");
  auto busy = [&]() {
    bool b = c->last_stream_valid && hipStreamQuery(c->last_stream) == hipErrorNotReady;
    for (hipStream_t ps : c->pull) b = b || (ps && hipStreamQuery(ps) == hipErrorNotReady);
    return b;
  };
  for (int spin = 0; spin < timeout_ms; ++spin) {
    if (!busy()) return 0;
    usleep(1000);
  }
  if (!busy()) return 0;
  c->aborted = true;
  if (c->flags_host) {
    volatile uint32_t* f = c->flags_host;
    const size_t nwait = (size_t)c->world * kSlots + (size_t)c->world * c->world;       // ready + done (the err words stay)
    for (int round = 0; round < 2000; ++round) {      // queued flag writes of this rank may overwrite the poison: repeat while draining (<= 2 s)
      for (size_t i = 0; i < nwait; ++i) f[i] = 0x7fffffffu;
      __sync_synchronize();
      if (!busy()) break;
      usleep(1000);
    }
  }
  return 1;
}

extern "C" void icv_ipc_destroy(icv_ipc* c) {
  if (!c) return;
  bool have_stream = false;
  for (hipStream_t s : c->pull) have_stream = have_stream || s;
  if (have_stream) (void)icv_ipc_drain(c, c->drain_ms);
  for (hipStream_t s : c->pull)
    if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
  if (c->arrived) (void)hipFree(c->arrived);
  for (hipEvent_t e : c->landed)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->started)
    if (e) (void)hipEventDestroy(e);
  for (void* b : c->peer_base)
    if (b) (void)hipIpcCloseMemHandle(b);
  if (c->heap && c->own_heap) (void)hipFree(c->heap);
  if (c->flags_host) {
    if (c->registered) (void)hipHostUnregister(c->flags_host);
    munmap(c->flags_host, c->flags_len);
  }
  delete c;
}
