// Native RCCL transport of libdg16 (include/dg16.h: dg16_rccl_*).
//
// What it replaces: the reference's mpc-net -- ProdNet's TCP/TLS star around the king (mpc-net/src/prod.rs) under
// the MpcNet trait (mpc-net/src/lib.rs:36-140) and the serialising MpcSerNet wrapper
// (dist-primitives/src/channel/mod.rs:8-57).  On one MI355X node the parties are the GPUs: payloads stay in HBM
// and move over xGMI as RCCL point-to-point transfers fused in one group per collective (xGMI is point to point --
// a gather to the king is n - 1 concurrent link transfers, an all-to-all uses all seven links of every GPU), ordered
// on the HIP stream the payload was produced on.  Nothing here synchronises the host.
//
// ONE COMMUNICATOR PER CHANNEL.  MpcNet's three MultiplexedStreamIDs (mpc-net/src/lib.rs:29-33) are independent
// streams: prove::C joins three d_msm on them (groth16/src/prove.rs:113-125) and the library's dg16_prove_c drives them
// from three host threads whose interleaving differs from party to party.  NCCL matches the point-to-point calls of
// one communicator in issue order and allows one host thread at a time per communicator -- so channel c owns
// communicator c (the first is formed by ncclCommInitRank and also serves the dg16_comm collectives, the other two are
// ncclCommSplit duplicates of it; a librccl without ncclCommSplit gets two fresh ids broadcast over the first), every
// call holds that communicator's mutex while it enqueues, and a payload of channel 1 can never meet a receive of
// channel 0 whatever the thread schedule is.
//
// librccl is bound at run time: a process that has torch loaded already holds torch's copy (same soname) and gets
// that one; a plain C consumer gets /opt/rocm's.  Linking it would make every libdg16 user load RCCL.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>

#include "ctx.h"

namespace {

// the part of rccl.h this file needs (ABI of NCCL 2.x, unchanged across RCCL releases)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;        // ncclSuccess == 0
constexpr int kNcclInt8 = 0;     // ncclInt8 / ncclChar

struct Api {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t*, void*) = nullptr;     // optional (NCCL >= 2.18)
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};

thread_local std::string g_err;

Api& api() {
  static Api a;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names)
      if ((a.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;        // the copy this process already uses
    if (!a.lib)
      for (const char* n : names)
        if ((a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!a.lib) {
      a.why = std::string("librccl not found: ") + dlerror();
      return;
    }
    auto sym = [&](const char* name) {
      void* f = dlsym(a.lib, name);
      if (!f && a.why.empty()) a.why = std::string("librccl lacks ") + name;
      return f;
    };
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.CommCount = (decltype(a.CommCount))sym("ncclCommCount");
    a.CommUserRank = (decltype(a.CommUserRank))sym("ncclCommUserRank");
    a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
    a.Send = (decltype(a.Send))sym("ncclSend");
    a.Recv = (decltype(a.Recv))sym("ncclRecv");
    a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
    a.Broadcast = (decltype(a.Broadcast))sym("ncclBroadcast");
    a.CommSplit = (decltype(a.CommSplit))dlsym(a.lib, "ncclCommSplit");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
  });
  return a;
}

bool api_ok() {
  Api& a = api();
  if (!a.lib || !a.why.empty()) {
    g_err = a.why;
    return false;
  }
  return true;
}

}  // namespace

constexpr int kChannels = 3;      // MultiplexedStreamID::{Zero, One, Two}

struct dg16_rccl {
  dg16_ctx* ctx = nullptr;
  ncclComm_t comm[kChannels] = {nullptr, nullptr, nullptr};   // [c]: channel c; [0] also: the dg16_comm collectives
  std::mutex mu[kChannels];                                   // one enqueuing host thread per communicator
  bool split = false;                                         // channels 1, 2 made by ncclCommSplit (else: fresh ids)
  unsigned n = 0, me = 0;
  dg16_comm comm_vt{};
  dg16_net net_vt{};
  bool check(ncclResult_t r, const char* what) {
    if (r == 0) return true;
    g_err = std::string(what) + ": " + (api().GetErrorString ? api().GetErrorString(r) : "rccl error");
    return false;
  }
};

namespace {

unsigned rc_n(void* self) { return ((dg16_rccl*)self)->n; }
unsigned rc_me(void* self) { return ((dg16_rccl*)self)->me; }
int rc_is_init(void* self) {
  auto* h = (dg16_rccl*)self;
  return h->comm[0] && h->comm[1] && h->comm[2];
}

// The enqueue of one call on one communicator: the device of the context, the communicator's mutex (NCCL allows one
// host thread at a time per communicator; the lock covers the enqueue only -- completion is stream-ordered).
struct Enq {
  dg16_rccl* h;
  ncclComm_t comm;
  std::unique_lock<std::mutex> lock;
  bool dev_ok;
  Enq(dg16_rccl* h_, int channel) : h(h_), comm(h_->comm[channel]), lock(h_->mu[channel]) {
    dev_ok = hipSetDevice(h->ctx->device) == hipSuccess;
  }
};
bool channel_ok(int channel) {
  if (channel >= 0 && channel < kChannels) return true;
  g_err = "channel must be 0, 1 or 2 (MultiplexedStreamID)";
  return false;
}

int rc_all_gather(void* self, const void* send, size_t bytes, void* recv, void* stream) {
  Enq e((dg16_rccl*)self, 0);
  if (!e.dev_ok) return DG16_ERR_HIP;
  return e.h->check(api().AllGather(send, recv, bytes, kNcclInt8, e.comm, (hipStream_t)stream), "ncclAllGather")
             ? DG16_OK
             : DG16_ERR_NET;
}

int rc_all_to_all(void* self, const void* send, void* recv, size_t bytes_per_peer, void* stream) {
  Enq e((dg16_rccl*)self, 0);
  auto* h = e.h;
  Api& a = api();
  if (!e.dev_ok) return DG16_ERR_HIP;
  bool ok = h->check(a.GroupStart(), "ncclGroupStart");
  for (unsigned p = 0; ok && p < h->n; p++) {
    ok = h->check(a.Send((const uint8_t*)send + p * bytes_per_peer, bytes_per_peer, kNcclInt8, (int)p, e.comm,
                         (hipStream_t)stream), "ncclSend") &&
         h->check(a.Recv((uint8_t*)recv + p * bytes_per_peer, bytes_per_peer, kNcclInt8, (int)p, e.comm,
                         (hipStream_t)stream), "ncclRecv");
  }
  // the group is always closed, also after a failed call inside it (an open group would swallow later collectives)
  const bool closed = h->check(a.GroupEnd(), "ncclGroupEnd");
  return ok && closed ? DG16_OK : DG16_ERR_NET;
}

// client_send_or_king_receive (mpc-net/src/lib.rs:61-99): n - 1 sends meet n - 1 receives on the king; the king's
// own block is a device copy on the same stream.  The calls of `channel` go to that channel's communicator: within it
// every party issues the collectives of a protocol in the same order (the d_* functions are straight-line code per
// channel), so issue-order matching is exact; across channels nothing is shared.
int rc_gather(void* self, int channel, const void* send, size_t bytes, void* recv, void* stream) {
  if (!channel_ok(channel)) return DG16_ERR_BAD_ARG;
  Enq e((dg16_rccl*)self, channel);
  auto* h = e.h;
  Api& a = api();
  hipStream_t s = (hipStream_t)stream;
  if (!e.dev_ok) return DG16_ERR_HIP;
  if (h->me != 0) return h->check(a.Send(send, bytes, kNcclInt8, 0, e.comm, s), "ncclSend") ? DG16_OK : DG16_ERR_NET;
  if (hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return DG16_ERR_HIP;
  bool ok = h->check(a.GroupStart(), "ncclGroupStart");
  for (unsigned p = 1; ok && p < h->n; p++)
    ok = h->check(a.Recv((uint8_t*)recv + p * bytes, bytes, kNcclInt8, (int)p, e.comm, s), "ncclRecv");
  const bool closed = h->check(a.GroupEnd(), "ncclGroupEnd");
  return ok && closed ? DG16_OK : DG16_ERR_NET;
}

// client_receive_or_king_send (mpc-net/src/lib.rs:102-140)
int rc_scatter(void* self, int channel, const void* send, size_t bytes, void* recv, void* stream) {
  if (!channel_ok(channel)) return DG16_ERR_BAD_ARG;
  Enq e((dg16_rccl*)self, channel);
  auto* h = e.h;
  Api& a = api();
  hipStream_t s = (hipStream_t)stream;
  if (!e.dev_ok) return DG16_ERR_HIP;
  if (h->me != 0) return h->check(a.Recv(recv, bytes, kNcclInt8, 0, e.comm, s), "ncclRecv") ? DG16_OK : DG16_ERR_NET;
  if (hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return DG16_ERR_HIP;
  bool ok = h->check(a.GroupStart(), "ncclGroupStart");
  for (unsigned p = 1; ok && p < h->n; p++)
    ok = h->check(a.Send((const uint8_t*)send + p * bytes, bytes, kNcclInt8, (int)p, e.comm, s), "ncclSend");
  const bool closed = h->check(a.GroupEnd(), "ncclGroupEnd");
  return ok && closed ? DG16_OK : DG16_ERR_NET;
}

int rc_send_to(void* self, unsigned peer, int channel, const void* send, size_t bytes, void* stream) {
  auto* h = (dg16_rccl*)self;
  if (peer >= h->n || peer == h->me || !channel_ok(channel)) return DG16_ERR_BAD_ARG;
  Enq e(h, channel);
  if (!e.dev_ok) return DG16_ERR_HIP;
  return h->check(api().Send(send, bytes, kNcclInt8, (int)peer, e.comm, (hipStream_t)stream), "ncclSend") ? DG16_OK
                                                                                                           : DG16_ERR_NET;
}
int rc_recv_from(void* self, unsigned peer, int channel, void* recv, size_t bytes, void* stream) {
  auto* h = (dg16_rccl*)self;
  if (peer >= h->n || peer == h->me || !channel_ok(channel)) return DG16_ERR_BAD_ARG;
  Enq e(h, channel);
  if (!e.dev_ok) return DG16_ERR_HIP;
  return h->check(api().Recv(recv, bytes, kNcclInt8, (int)peer, e.comm, (hipStream_t)stream), "ncclRecv") ? DG16_OK
                                                                                                           : DG16_ERR_NET;
}

// Communicators of channels 1 and 2.  ncclCommSplit(color 0, key = rank) duplicates the first communicator (a
// collective over it, no out-of-band exchange).  Without that entry point (or with DG16_RCCL_NO_SPLIT=1, which the
// tests use to run this branch): rank 0 makes two fresh ids, broadcasts them over the first communicator and every
// rank joins them with ncclCommInitRank.
bool make_channel_comms(dg16_rccl* h) {
  Api& a = api();
  const char* no_split = getenv("DG16_RCCL_NO_SPLIT");
  if (a.CommSplit && !(no_split && no_split[0] == '1')) {
    for (int c = 1; c < kChannels; c++)
      if (!h->check(a.CommSplit(h->comm[0], 0, (int)h->me, &h->comm[c], nullptr), "ncclCommSplit")) return false;
    h->split = true;
    return true;
  }
  constexpr size_t kIds = (kChannels - 1) * sizeof(ncclUniqueId);
  ncclUniqueId ids[kChannels - 1];
  if (h->me == 0)
    for (auto& id : ids)
      if (!h->check(a.GetUniqueId(&id), "ncclGetUniqueId")) return false;
  void* dbuf = nullptr;
  if (hipMalloc(&dbuf, kIds) != hipSuccess) {
    g_err = "hipMalloc (channel ids)";
    return false;
  }
  // on a private NON-BLOCKING stream: the NULL stream would synchronise with every blocking stream of the process
  hipStream_t bs = nullptr;
  bool ok = hipStreamCreateWithFlags(&bs, hipStreamNonBlocking) == hipSuccess &&
            hipMemcpyAsync(dbuf, ids, kIds, hipMemcpyHostToDevice, bs) == hipSuccess &&
            h->check(a.Broadcast(dbuf, dbuf, kIds, kNcclInt8, 0, h->comm[0], bs), "ncclBroadcast") &&
            hipMemcpyAsync(ids, dbuf, kIds, hipMemcpyDeviceToHost, bs) == hipSuccess &&
            hipStreamSynchronize(bs) == hipSuccess;
  if (bs) hipStreamDestroy(bs);
  hipFree(dbuf);
  if (!ok) {
    if (g_err.empty()) g_err = "broadcast of the channel ids failed";
    return false;
  }
  for (int c = 1; c < kChannels; c++)
    if (!h->check(a.CommInitRank(&h->comm[c], (int)h->n, ids[c - 1], (int)h->me), "ncclCommInitRank (channel)"))
      return false;
  return true;
}

void destroy_comms(dg16_rccl* h) {
  for (int c = kChannels - 1; c >= 0; c--)
    if (h->comm[c]) {
      api().CommDestroy(h->comm[c]);
      h->comm[c] = nullptr;
    }
}

}  // namespace

namespace dg16 {
// Is `net` the vtable of a dg16_rccl handle (dg16_rccl_net)?  dg16_prove_c orders its three d_msm on such a net:
// RCCL guarantees progress of communicators used CONCURRENTLY on one device only if all their kernels can be co-resident
// and the ranks enqueue on them in a consistent order -- three host threads per rank, each enqueueing on its own
// communicator in an order that differs from rank to rank, is the pattern NCCL documents as a deadlock hazard.
bool net_is_rccl(const dg16_net* net) { return net && net->gather_to_king == &rc_gather; }
}  // namespace dg16

extern "C" {

const char* dg16_rccl_error(void) { return g_err.c_str(); }

int dg16_rccl_unique_id(void* out128) {
  if (!out128) return DG16_ERR_BAD_ARG;
  if (!api_ok()) return DG16_ERR_UNSUPPORTED;
  ncclUniqueId id;
  ncclResult_t r = api().GetUniqueId(&id);
  if (r != 0) {
    g_err = std::string("ncclGetUniqueId: ") + api().GetErrorString(r);
    return DG16_ERR_NET;
  }
  memcpy(out128, id.internal, sizeof(id.internal));
  return DG16_OK;
}

int dg16_rccl_create(dg16_ctx* ctx, const void* unique_id128, unsigned n_ranks, unsigned rank, dg16_rccl** out) {
  if (!ctx || !unique_id128 || !out || n_ranks == 0 || rank >= n_ranks) return DG16_ERR_BAD_ARG;
  *out = nullptr;
  if (!api_ok()) return DG16_ERR_UNSUPPORTED;
  if (hipSetDevice(ctx->device) != hipSuccess) return DG16_ERR_HIP;
  auto* h = new dg16_rccl();
  h->ctx = ctx;
  h->n = n_ranks;
  h->me = rank;
  ncclUniqueId id;
  memcpy(id.internal, unique_id128, sizeof(id.internal));
  if (!h->check(api().CommInitRank(&h->comm[0], (int)n_ranks, id, (int)rank), "ncclCommInitRank") ||
      !make_channel_comms(h)) {
    destroy_comms(h);
    delete h;
    return DG16_ERR_NET;
  }
  h->comm_vt = dg16_comm{h, rc_n, rc_me, rc_all_gather, rc_all_to_all};
  h->net_vt = dg16_net{h, rc_n, rc_me, rc_gather, rc_scatter, rc_is_init, rc_send_to, rc_recv_from};
  *out = h;
  return DG16_OK;
}

// what the COMMUNICATORS report (ncclCommCount / ncclCommUserRank), not what the caller asked for: all three must agree
int dg16_rccl_ranks(dg16_rccl* h, unsigned* n_ranks, unsigned* rank) {
  if (!h || !h->comm[0]) return DG16_ERR_BAD_ARG;
  int n0 = 0, me0 = 0;
  for (int c = 0; c < kChannels; c++) {
    int n = 0, me = 0;
    if (!h->comm[c] || !h->check(api().CommCount(h->comm[c], &n), "ncclCommCount") ||
        !h->check(api().CommUserRank(h->comm[c], &me), "ncclCommUserRank"))
      return DG16_ERR_NET;
    if (c == 0) {
      n0 = n;
      me0 = me;
    } else if (n != n0 || me != me0) {
      g_err = "channel communicators disagree on the rank layout";
      return DG16_ERR_NET;
    }
  }
  if (n_ranks) *n_ranks = (unsigned)n0;
  if (rank) *rank = (unsigned)me0;
  return DG16_OK;
}

/* 1 if channels 1 and 2 are ncclCommSplit duplicates of the first communicator, 0 if they were joined through fresh
 * broadcast ids */
int dg16_rccl_channels_split(dg16_rccl* h) { return h && h->split ? 1 : 0; }

const dg16_comm* dg16_rccl_comm(dg16_rccl* h) { return h ? &h->comm_vt : nullptr; }
const dg16_net* dg16_rccl_net(dg16_rccl* h) { return h ? &h->net_vt : nullptr; }

void dg16_rccl_destroy(dg16_rccl* h) {
  if (!h) return;
  hipSetDevice(h->ctx->device);
  hipDeviceSynchronize();
  destroy_comms(h);
  delete h;
}

}  // extern "C"
