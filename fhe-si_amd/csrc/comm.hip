// comm.hip -- the multi-GPU side of the C ABI: RCCL collectives over xGMI on the library's own streams.
//
// The data-parallel model has ONE collective in its set-up (the key-switch matrix generated on one rank is broadcast into every rank's
// HBM replica: KeySwitchSI::keySwitchMatrix, FHE-SI.cpp:206-208) and, for wave-scheduled Matrix<Ciphertext> arithmetic whose waves
// are sharded over the ranks, one exchange of each wave's outputs (the next wave reads ciphertexts produced by every rank) plus the
// optional all-reduce of partial scaled-up sums (Ciphertext.cpp:135-142: scaled-up addition is linear).  No torch here: librccl is
// resolved at run time (dlopen) so that single-GPU users need no RCCL at all, and communicators are plain ncclComm_t values made by
// the caller (process per GPU: ncclCommInitRank) or by fhesi_comm_init_all (one process, one host thread per GPU).
//
// "Loopback" groups exist for ONE purpose: ranks that share a device (RCCL refuses duplicate GPUs), i.e. exercising the N > 1 host
// logic on a single-GPU box.  They move the same bytes with device-to-device copies between the ranks' buffers and host barriers.
#include "../../include/fhesi_hip.h"
#include "fhesi_internal.h"

#include <dlfcn.h>

#include <condition_variable>
#include <mutex>

namespace {

typedef void* nccl_comm_t;
struct RcclApi {
  void* lib = nullptr;
  int (*CommInitAll)(nccl_comm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(nccl_comm_t) = nullptr;
  int (*CommCount)(nccl_comm_t, int*) = nullptr;
  int (*CommUserRank)(nccl_comm_t, int*) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
const int kNcclUint64 = 5, kNcclSum = 0;     // ncclDataType_t / ncclRedOp_t values of rccl.h

RcclApi* rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (!api.lib) return;
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(api.lib, n); if (!p) ok = false; return p; };
    api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.CommCount = (decltype(api.CommCount))sym("ncclCommCount");
    api.CommUserRank = (decltype(api.CommUserRank))sym("ncclCommUserRank");
    api.Broadcast = (decltype(api.Broadcast))sym("ncclBroadcast");
    api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { dlclose(api.lib); api.lib = nullptr; }
  });
  return api.lib ? &api : nullptr;
}

// ranks of one process that share a device: publish buffer pointers, meet at a host barrier, copy.  A rank whose part of a collective
// fails still takes every barrier of that collective (its peers would wait for ever otherwise) and records the failure; after the last
// barrier every rank of the group returns an error for that collective.
struct LoopGroup {
  int nranks = 0, refs = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  long generation = 0;
  std::atomic<long> failed_at{-1};     // generation of the closing barrier of a collective in which some rank failed
  std::vector<void*> ptr;
  std::vector<hipEvent_t> ev;          // exchange_begin: "my shard is complete" on each rank's compute stream
  long barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const long gen = generation;
    if (++arrived == nranks) { arrived = 0; ++generation; cv.notify_all(); }
    else cv.wait(lk, [&] { return generation != gen; });
    return gen;
  }
  // closing barrier of a collective: rc = this rank's status; returns non-zero on EVERY rank when any rank failed
  int close(int rc) {
    long gen_now;
    { std::lock_guard<std::mutex> lk(mu); gen_now = generation; }
    if (rc) failed_at.store(gen_now);
    const long gen = barrier();
    if (!rc && failed_at.load() == gen) { fhesi_set_error("loopback collective: another rank of the group failed"); return 1; }
    return rc;
  }
};
#define HIP_SOFT(rc, expr) do { if (!(rc)) { hipError_t e__ = (expr); if (e__ != hipSuccess) { fhesi_set_error("%s failed: %s", #expr, hipGetErrorString(e__)); (rc) = 1; } } } while (0)

}  // namespace

struct fhesi_comm {
  int rank = 0, nranks = 1;
  nccl_comm_t nccl = nullptr;          // RCCL communicator (owned when made by fhesi_comm_init_all)
  bool owns_nccl = false;
  LoopGroup* loop = nullptr;           // loopback group (ranks sharing a device)
  // fhesi_comm_exchange_begin / _end: the exchanges run on a stream of the communicator's own, behind an event of the compute stream
  hipStream_t xs = nullptr;
  hipEvent_t ready = nullptr;
  int xs_device = -1;
  int pending = 0;                     // exchanges begun and not yet ended
};

#define RCCL_TRY(api, expr) do { int r__ = (expr); if (r__ != 0) { fhesi_set_error("%s failed: %s", #expr, (api)->GetErrorString ? (api)->GetErrorString(r__) : "RCCL error"); return 1; } } while (0)

extern "C" int fhesi_comm_init_all(int32_t ndev, const int32_t* devices, fhesi_comm** comms_out) {
  if (ndev < 1 || !devices || !comms_out) FHESI_FAIL("comm_init_all: bad arguments");
  bool distinct = true;
  for (int i = 0; i < ndev; ++i) for (int j = 0; j < i; ++j) distinct = distinct && devices[i] != devices[j];
  if (!distinct) {
    // ranks sharing a device: RCCL rejects duplicate GPUs, so this group moves its bytes with device copies (plumbing checks on a 1-GPU box)
    LoopGroup* g = new LoopGroup();
    g->nranks = ndev; g->refs = ndev; g->ptr.assign(ndev, nullptr); g->ev.assign(ndev, nullptr);
    for (int i = 0; i < ndev; ++i) { fhesi_comm* c = new fhesi_comm(); c->rank = i; c->nranks = ndev; c->loop = g; comms_out[i] = c; }
    return 0;
  }
  RcclApi* api = rccl();
  if (!api) FHESI_FAIL("comm_init_all: librccl.so.1 could not be loaded");
  std::vector<nccl_comm_t> cs(ndev);
  std::vector<int> devs(devices, devices + ndev);
  RCCL_TRY(api, api->CommInitAll(cs.data(), ndev, devs.data()));
  for (int i = 0; i < ndev; ++i) { fhesi_comm* c = new fhesi_comm(); c->rank = i; c->nranks = ndev; c->nccl = cs[i]; c->owns_nccl = true; comms_out[i] = c; }
  return 0;
}

extern "C" int fhesi_comm_from_rccl(void* nccl_comm, fhesi_comm** out) {
  if (!nccl_comm || !out) FHESI_FAIL("comm_from_rccl: null argument");
  RcclApi* api = rccl();
  if (!api) FHESI_FAIL("comm_from_rccl: librccl.so.1 could not be loaded");
  fhesi_comm* c = new fhesi_comm();
  c->nccl = nccl_comm;
  if (api->CommCount(nccl_comm, &c->nranks) != 0 || api->CommUserRank(nccl_comm, &c->rank) != 0) { delete c; FHESI_FAIL("comm_from_rccl: not a usable ncclComm_t"); }
  *out = c;
  return 0;
}

extern "C" int fhesi_comm_destroy(fhesi_comm* c) {
  if (!c) return 0;
  if (c->xs) { hipSetDevice(c->xs_device); hipStreamSynchronize(c->xs); hipStreamDestroy(c->xs); }
  if (c->ready) hipEventDestroy(c->ready);
  if (c->nccl && c->owns_nccl) { RcclApi* api = rccl(); if (api) api->CommDestroy(c->nccl); }
  if (c->loop) { bool last; { std::lock_guard<std::mutex> lk(c->loop->mu); last = --c->loop->refs == 0; } if (last) delete c->loop; }
  delete c;
  return 0;
}
extern "C" int32_t fhesi_comm_rank(const fhesi_comm* c) { return c ? c->rank : -1; }
extern "C" int32_t fhesi_comm_size(const fhesi_comm* c) { return c ? c->nranks : 0; }

// buf (bytes) of rank `root` into every rank's buf, on the context's stream
static int comm_broadcast(fhesi_ctx* ctx, fhesi_comm* c, void* buf, size_t bytes, int root) {
  if (root < 0 || root >= c->nranks) FHESI_FAIL("broadcast: root %d outside the %d ranks", root, c->nranks);
  if (c->nranks == 1 && !c->nccl) return 0;       // (a one-rank RCCL communicator still goes through RCCL)
  if (c->loop) {
    LoopGroup* g = c->loop;
    int rc = 0;
    g->ptr[c->rank] = buf;
    g->barrier();
    if (c->rank != root) HIP_SOFT(rc, hipMemcpyAsync(buf, g->ptr[root], bytes, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_SOFT(rc, hipStreamSynchronize(ctx->stream));
    return g->close(rc);                       // the root's buffer stays untouched until every rank has read it
  }
  RcclApi* api = rccl();
  if (!api) FHESI_FAIL("broadcast: librccl.so.1 could not be loaded");
  if (bytes % 8) FHESI_FAIL("broadcast: byte count must be a multiple of 8");
  RCCL_TRY(api, api->Broadcast(buf, buf, bytes / 8, kNcclUint64, root, c->nccl, ctx->stream));
  return 0;
}

// The set-up collective: rank `root`'s key-switch matrix into every rank's replica (KeySwitchSI::keySwitchMatrix rows, FHE-SI.cpp:206-208)
extern "C" int fhesi_ksk_broadcast(fhesi_ksk* k, fhesi_comm* comm, int32_t root) {
  if (!k || !comm) FHESI_FAIL("ksk_broadcast: null argument");
  fhesi_ctx* ctx = k->ctx;
  HIP_TRY(hipSetDevice(ctx->device));
  FHESI_TRY(comm_broadcast(ctx, comm, k->d_rows, k->bytes, root));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (comm->rank != root) k->aux_valid = false;      // the rows changed: derived tables are rebuilt at the next key switch
  return 0;
}

extern "C" int fhesi_comm_broadcast_dev(fhesi_ctx* ctx, fhesi_comm* comm, void* buf_dev, size_t bytes, int32_t root) {
  if (!ctx || !comm) FHESI_FAIL("broadcast: null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  FHESI_TRY(comm_broadcast(ctx, comm, buf_dev, bytes, root));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return 0;
}

// Exchange of sharded results: rank r holds words [offsets[r], offsets[r+1]) of base_dev; afterwards every rank holds all of them.
// One (grouped) broadcast per producing rank: shards differ in size and a wave has few of them, so this beats a padded all-gather.
extern "C" int fhesi_comm_exchange(fhesi_ctx* ctx, fhesi_comm* comm, uint64_t* base_dev, const int64_t* offsets_words) {
  if (!ctx || !comm || !offsets_words) FHESI_FAIL("exchange: null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  if (comm->nranks == 1 && !comm->nccl) return 0;
  if (comm->loop) {
    LoopGroup* g = comm->loop;
    int rc = 0;
    g->ptr[comm->rank] = base_dev;
    HIP_SOFT(rc, hipStreamSynchronize(ctx->stream));  // my shard is complete before anyone reads it
    g->barrier();
    for (int r = 0; r < comm->nranks && !rc; ++r) {
      const i64 lo = offsets_words[r], hi = offsets_words[r + 1];
      if (r == comm->rank || hi <= lo) continue;
      HIP_SOFT(rc, hipMemcpyAsync(base_dev + lo, (const u64*)g->ptr[r] + lo, (size_t)(hi - lo) * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    HIP_SOFT(rc, hipStreamSynchronize(ctx->stream));
    return g->close(rc);
  }
  RcclApi* api = rccl();
  if (!api) FHESI_FAIL("exchange: librccl.so.1 could not be loaded");
  {
    // equal shards (the usual case: a wave's groups divide evenly over the ranks) are ONE in-place all-gather: a single ring over xGMI
    // instead of one broadcast per producing rank
    const i64 each = offsets_words[1] - offsets_words[0];
    bool uniform = each > 0;
    for (int r = 0; r < comm->nranks && uniform; ++r) uniform = offsets_words[r + 1] - offsets_words[r] == each;
    if (uniform) {
      RCCL_TRY(api, api->AllGather(base_dev + offsets_words[comm->rank], base_dev + offsets_words[0], (size_t)each, kNcclUint64, comm->nccl, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      return 0;
    }
  }
  RCCL_TRY(api, api->GroupStart());
  for (int r = 0; r < comm->nranks; ++r) {
    const i64 lo = offsets_words[r], hi = offsets_words[r + 1];
    if (hi <= lo) continue;
    RCCL_TRY(api, api->Broadcast(base_dev + lo, base_dev + lo, (size_t)(hi - lo), kNcclUint64, r, comm->nccl, ctx->stream));
  }
  RCCL_TRY(api, api->GroupEnd());
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return 0;
}

// The same exchange in two halves, so that a rank can go on computing while its finished shard travels (the waves of Regression::Regress,
// Regression.h:102-149 over Matrix.cpp:182-263: a wave's outputs are needed by the NEXT wave only, so a wave cut into chunks can exchange
// chunk k while chunk k + 1 is computed).
//   begin: the exchange of [offsets[0], offsets[G]) is ENQUEUED on the communicator's own stream behind everything already on the
//          context's stream (an event, no host wait); the context's stream is free for further launches at once.  Every rank of the
//          group calls begin for the same exchanges in the same order.
//   end:   returns when every exchange begun since the last end has landed in THIS rank's buffer (and, in a loopback group, in every
//          rank's); later work on the context's stream may read all of it.
static int comm_xs(fhesi_ctx* ctx, fhesi_comm* c) {
  if (c->xs) return 0;
  HIP_TRY(hipStreamCreateWithFlags(&c->xs, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
  c->xs_device = ctx->device;
  return 0;
}
extern "C" int fhesi_comm_exchange_begin(fhesi_ctx* ctx, fhesi_comm* comm, uint64_t* base_dev, const int64_t* offsets_words) {
  if (!ctx || !comm || !offsets_words) FHESI_FAIL("exchange_begin: null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  if (comm->nranks == 1 && !comm->nccl) return 0;
  FHESI_TRY(comm_xs(ctx, comm));
  HIP_TRY(hipEventRecord(comm->ready, ctx->stream));          // my shard is complete when this event fires
  if (comm->loop) {
    LoopGroup* g = comm->loop;
    int rc = 0;
    g->ptr[comm->rank] = base_dev;
    g->ev[comm->rank] = comm->ready;
    g->barrier();                                               // (host side only: nobody waits for the GPU here)
    for (int r = 0; r < comm->nranks && !rc; ++r) {
      const i64 lo = offsets_words[r], hi = offsets_words[r + 1];
      if (r == comm->rank || hi <= lo) continue;
      HIP_SOFT(rc, hipStreamWaitEvent(comm->xs, g->ev[r], 0));  // rank r's shard is complete before it is read
      HIP_SOFT(rc, hipMemcpyAsync(base_dev + lo, (const u64*)g->ptr[r] + lo, (size_t)(hi - lo) * 8, hipMemcpyDeviceToDevice, comm->xs));
    }
    ++comm->pending;
    return g->close(rc);                                        // every wait is enqueued: the events may be recorded again
  }
  RcclApi* api = rccl();
  if (!api) FHESI_FAIL("exchange_begin: librccl.so.1 could not be loaded");
  HIP_TRY(hipStreamWaitEvent(comm->xs, comm->ready, 0));
  const i64 each = offsets_words[1] - offsets_words[0];
  bool uniform = each > 0;
  for (int r = 0; r < comm->nranks && uniform; ++r) uniform = offsets_words[r + 1] - offsets_words[r] == each;
  ++comm->pending;
  if (uniform) {
    RCCL_TRY(api, api->AllGather(base_dev + offsets_words[comm->rank], base_dev + offsets_words[0], (size_t)each, kNcclUint64, comm->nccl, comm->xs));
    return 0;
  }
  RCCL_TRY(api, api->GroupStart());
  for (int r = 0; r < comm->nranks; ++r) {
    const i64 lo = offsets_words[r], hi = offsets_words[r + 1];
    if (hi <= lo) continue;
    RCCL_TRY(api, api->Broadcast(base_dev + lo, base_dev + lo, (size_t)(hi - lo), kNcclUint64, r, comm->nccl, comm->xs));
  }
  RCCL_TRY(api, api->GroupEnd());
  return 0;
}
extern "C" int fhesi_comm_exchange_end(fhesi_ctx* ctx, fhesi_comm* comm) {
  if (!ctx || !comm) FHESI_FAIL("exchange_end: null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  if (!comm->pending) return 0;
  comm->pending = 0;
  if (comm->loop) {
    int rc = 0;
    HIP_SOFT(rc, hipStreamSynchronize(comm->xs));
    return comm->loop->close(rc);            // every rank has read every shard: the buffers may be reallocated
  }
  HIP_TRY(hipStreamSynchronize(comm->xs));
  return 0;
}

// rows_dev[i] <- (sum over ranks of rows_dev[i]) mod q_{slot(i)}: the exact accumulator of partial scaled-up sums held by the ranks
// (Ciphertext.cpp:135-142).  Residues are below 2^60, so the sum of up to 16 ranks fits a word before the reduction.
__global__ void __launch_bounds__(256) reduce_after_sum_kernel(u64* __restrict__ rows, i64 n, int L, const PrimeConst* __restrict__ pcs) {
  const i64 row = blockIdx.y;
  const PrimeConst pc = pcs[row % L];
  u64* x = rows + row * n;
  for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (i64)gridDim.x * blockDim.x) {
    const u64 v = x[j];
    x[j] = d_shoup(v, 1, pc.one_sh, pc.q);
  }
}
extern "C" int fhesi_comm_allreduce_rows(fhesi_ctx* ctx, fhesi_comm* comm, uint64_t* rows_dev, int64_t count) {
  if (!ctx || !comm) FHESI_FAIL("allreduce: null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  if ((comm->nranks == 1 && !comm->nccl) || !count) return 0;
  if (comm->nranks > 16) FHESI_FAIL("allreduce: more than 16 ranks would overflow the 64-bit partial sums");
  const i64 n = ctx->phim, words = count * ctx->L * n;
  if (comm->loop) {
    // sum through a staging copy of every other rank's rows (plumbing path: ranks share the device)
    LoopGroup* g = comm->loop;
    void* stage = nullptr;
    int rc = 0;
    HIP_SOFT(rc, hipMalloc(&stage, (size_t)words * 8));
    HIP_SOFT(rc, hipMemcpyAsync(stage, rows_dev, (size_t)words * 8, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_SOFT(rc, hipStreamSynchronize(ctx->stream));
    g->ptr[comm->rank] = rc ? nullptr : stage;
    g->barrier();
    for (int r = 0; r < comm->nranks && !rc; ++r) {
      if (r == comm->rank) continue;
      if (!g->ptr[r]) { fhesi_set_error("allreduce: rank %d has no staging copy", r); rc = 1; break; }
      rc = launch_ew_op(ctx, rows_dev, (const u64*)g->ptr[r], count, ctx->L, nullptr, FHESI_OP_ADD);
    }
    HIP_SOFT(rc, hipStreamSynchronize(ctx->stream));
    rc = g->close(rc);                         // every rank has read every staging copy
    if (stage) hipFree(stage);
    return rc;
  }
  RcclApi* api = rccl();
  if (!api) FHESI_FAIL("allreduce: librccl.so.1 could not be loaded");
  RCCL_TRY(api, api->AllReduce(rows_dev, rows_dev, (size_t)words, kNcclUint64, kNcclSum, comm->nccl, ctx->stream));
  unsigned gx = (unsigned)((n + 255) / 256);
  if (gx > 64) gx = 64;
  reduce_after_sum_kernel<<<dim3(gx, (unsigned)(count * ctx->L)), 256, 0, ctx->stream>>>(rows_dev, n, ctx->L, ctx->d_pc);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return 0;
}
