// Host side of the dist-primitives mirror (see dist_impl.h): PackedSharingParams, the MpcNet-shaped
// transport with an in-process LocalNet, and d_fft / d_ifft / d_msm / d_pp / deg_red / ext_wit::h with
// the reference's names, argument meaning and error behaviour.
#include <chrono>
#include <thread>

#include "dist_impl.h"

using namespace dg16;

// ---- transport -----------------------------------------------------------------------------------
// In-process n-party net (LocalTestNet, mpc-net/src/multi.rs:227-329): parties are host threads of one
// process, payloads are device buffers, the king reads the other parties' buffers directly.
struct dg16_localnet {
  unsigned n;
  std::mutex mu;
  std::condition_variable cv;
  struct Slot {
    unsigned arrived = 0, generation = 0;
    const void* send[kMaxParties] = {};
    void* king_recv = nullptr;
    const void* king_send = nullptr;
    size_t bytes = 0;                   // the king's length
    size_t party_bytes[kMaxParties] = {};   // what every party announced (mpc-net/src/lib.rs:116-124 checks them)
    bool failed = false;
    bool result_ok = true;              // outcome of the last completed round, read by the clients
  } slot[kChannels][2];   // [channel][0 = gather, 1 = scatter]
  // point-to-point mailboxes (send_to / recv_from, mpc-net/src/lib.rs:48-58): [channel][from][to]
  struct Box {
    const void* src = nullptr;
    size_t bytes = 0;
    bool posted = false, done = false, ok = true;
  } box[kChannels][kMaxParties][kMaxParties];
  dg16_net party[kMaxParties];
  struct PartyRef { dg16_localnet* net; unsigned id; } ref[kMaxParties];
  bool aborted = false;       // a party died: every pending and future collective fails ("Stream died")
  unsigned timeout_s = 120;
};

namespace dg16 {

// Rendezvous of all n parties on one slot; `king_work` runs on the king once everybody has arrived,
// everybody leaves after it is done.  Returns false if any party reported mismatching sizes.
template <class Fn>
static bool rendezvous(dg16_localnet* ln, dg16_localnet::Slot& s, unsigned id, Fn&& king_work) {
  std::unique_lock<std::mutex> lk(ln->mu);
  if (ln->aborted) return false;
  unsigned gen = s.generation;
  s.arrived++;
  const auto limit = std::chrono::seconds(ln->timeout_s);
  if (id == 0) {
    if (!ln->cv.wait_for(lk, limit, [&] { return s.arrived == ln->n || ln->aborted; }) || ln->aborted) {
      ln->aborted = true;     // a peer never arrived: fail everybody instead of hanging
      s.arrived = 0;          // ... and leave the slot clean for the next round after a reset
      ln->cv.notify_all();
      return false;
    }
    // unequal lengths are an error on every party, like the reference's length check (a shorter buffer would be
    // read or written past its end by the king's copies)
    bool ok = true;
    for (unsigned p = 0; p < ln->n; p++) ok = ok && s.party_bytes[p] == s.bytes;
    if (ok) {
      lk.unlock();
      ok = king_work();
      lk.lock();
    }
    s.result_ok = ok;
    s.arrived = 0;
    s.generation++;
    ln->cv.notify_all();
    return ok;
  }
  ln->cv.notify_all();
  if (!ln->cv.wait_for(lk, limit, [&] { return s.generation != gen || ln->aborted; }) || ln->aborted) {
    ln->aborted = true;
    s.arrived = 0;
    ln->cv.notify_all();
    return false;
  }
  return s.result_ok;
}

static int localnet_gather(void* self, int channel, const void* send_dev, size_t bytes, void* recv_dev, void* stream) {
  auto* ref = (dg16_localnet::PartyRef*)self;
  dg16_localnet* ln = ref->net;
  auto& s = ln->slot[channel][0];
  if (stream && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return DG16_ERR_HIP;  // payload complete
  {
    std::lock_guard<std::mutex> g(ln->mu);
    s.send[ref->id] = send_dev;
    s.party_bytes[ref->id] = bytes;
    if (ref->id == 0) { s.king_recv = recv_dev; s.bytes = bytes; }
  }
  bool ok = rendezvous(ln, s, ref->id, [&] {
    for (unsigned p = 0; p < ln->n; p++)
      if (hipMemcpy((uint8_t*)s.king_recv + p * s.bytes, s.send[p], s.bytes, hipMemcpyDeviceToDevice) != hipSuccess)
        return false;
    return hipDeviceSynchronize() == hipSuccess;   // device-to-device hipMemcpy may return early
  });
  return ok ? DG16_OK : DG16_ERR_NET;
}

static int localnet_scatter(void* self, int channel, const void* send_dev, size_t bytes, void* recv_dev, void* stream) {
  auto* ref = (dg16_localnet::PartyRef*)self;
  dg16_localnet* ln = ref->net;
  auto& s = ln->slot[channel][1];
  if (stream && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return DG16_ERR_HIP;
  {
    std::lock_guard<std::mutex> g(ln->mu);
    s.send[ref->id] = recv_dev;   // reuse the pointer table for the receive buffers
    s.party_bytes[ref->id] = bytes;
    if (ref->id == 0) { s.king_send = send_dev; s.bytes = bytes; }
  }
  bool ok = rendezvous(ln, s, ref->id, [&] {
    // the king checks that all outgoing buffers have the same length by construction (one `bytes`),
    // mpc-net/src/lib.rs:116-124
    for (unsigned p = 0; p < ln->n; p++)
      if (hipMemcpy((void*)s.send[p], (const uint8_t*)s.king_send + p * s.bytes, s.bytes, hipMemcpyDeviceToDevice) !=
          hipSuccess)
        return false;
    return hipDeviceSynchronize() == hipSuccess;
  });
  return ok ? DG16_OK : DG16_ERR_NET;
}
// send_to: post the device buffer, wait until the peer's recv_from has copied it (the reference's send completes
// when the frame is written; here the buffer must stay valid until it is read, so the sender waits for the copy)
static int localnet_send_to(void* self, unsigned peer, int channel, const void* send_dev, size_t bytes, void* stream) {
  auto* ref = (dg16_localnet::PartyRef*)self;
  dg16_localnet* ln = ref->net;
  if (peer >= ln->n || peer == ref->id || channel < 0 || channel >= kChannels) return DG16_ERR_BAD_ARG;
  if (stream && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return DG16_ERR_HIP;
  auto& b = ln->box[channel][ref->id][peer];
  std::unique_lock<std::mutex> lk(ln->mu);
  const auto limit = std::chrono::seconds(ln->timeout_s);
  auto fail = [&] { ln->aborted = true; b = {}; ln->cv.notify_all(); return DG16_ERR_NET; };
  if (!ln->cv.wait_for(lk, limit, [&] { return !b.posted || ln->aborted; }) || ln->aborted) return fail();
  b.src = send_dev;
  b.bytes = bytes;
  b.posted = true;
  b.done = false;
  ln->cv.notify_all();
  if (!ln->cv.wait_for(lk, limit, [&] { return b.done || ln->aborted; }) || ln->aborted) return fail();
  const bool ok = b.ok;
  b = {};
  ln->cv.notify_all();
  return ok ? DG16_OK : DG16_ERR_NET;
}
static int localnet_recv_from(void* self, unsigned peer, int channel, void* recv_dev, size_t bytes, void* stream) {
  auto* ref = (dg16_localnet::PartyRef*)self;
  dg16_localnet* ln = ref->net;
  if (peer >= ln->n || peer == ref->id || channel < 0 || channel >= kChannels) return DG16_ERR_BAD_ARG;
  if (stream && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return DG16_ERR_HIP;
  auto& b = ln->box[channel][peer][ref->id];
  std::unique_lock<std::mutex> lk(ln->mu);
  const auto limit = std::chrono::seconds(ln->timeout_s);
  if (!ln->cv.wait_for(lk, limit, [&] { return (b.posted && !b.done) || ln->aborted; }) || ln->aborted) {
    ln->aborted = true;
    ln->cv.notify_all();
    return DG16_ERR_NET;
  }
  bool ok = b.bytes == bytes;      // a frame of another length is a protocol error on both sides
  const void* src = b.src;
  if (ok) {
    lk.unlock();
    ok = hipMemcpy(recv_dev, src, bytes, hipMemcpyDeviceToDevice) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    lk.lock();
  }
  b.ok = ok;
  b.done = true;
  ln->cv.notify_all();
  return ok ? DG16_OK : DG16_ERR_NET;
}
static int localnet_is_init(void* self) { return !((dg16_localnet::PartyRef*)self)->net->aborted; }
static unsigned localnet_n(void* self) { return ((dg16_localnet::PartyRef*)self)->net->n; }
static unsigned localnet_id(void* self) { return ((dg16_localnet::PartyRef*)self)->id; }

static void net_check(int rc) {
  if (rc != DG16_OK) throw StatusError{DG16_ERR_NET, "MpcNet collective failed (Stream died / NotConnected)"};
}

// ---- typed implementation -----------------------------------------------------------------------------
template <class Fr>
struct Dist {
  static Fr* mat(const dg16_pss* pp, int which) {  // 0 pack [n][l], 1 unpack [l][n], 2 unpack2 [l][n]
    Fr* m = (Fr*)pp->mats;
    return which == 0 ? m : which == 1 ? m + pp->n * pp->l : m + 2 * pp->n * pp->l;
  }
  static Fr* mat_canon(const dg16_pss* pp, int which) { return mat(pp, which) + 3 * pp->n * pp->l; }
  static Fr* v2sum(const dg16_pss* pp) { return (Fr*)pp->mats + 6 * pp->n * pp->l; }

  // fft2_with_rearrange_pad (dfft/mod.rs:185-256).  px: this party's m/l elements (device).
  static void fft2_with_rearrange_pad(Call& k, const dg16_pss* pp, const dg16_net* net, int curve, Fr* px,
                                      size_t mbyl, bool rearrange, unsigned pad, bool degree2, unsigned log_m,
                                      int inverse, Fr* out, int sid) {
    const unsigned n = pp->n, l = pp->l;
    const size_t m = mbyl * l, total = (size_t)pad * m, out_per_party = total / l;
    const bool king = net->party_id(net->self) == 0;
    Fr* gathered = king ? (Fr*)ws(k.c, 18, (size_t)n * mbyl * sizeof(Fr)) : nullptr;
    net_check(net->gather_to_king(net->self, sid, px, mbyl * sizeof(Fr), gathered, k.s()));
    Fr* send = nullptr;
    if (king) {
      Fr* s1 = (Fr*)ws(k.c, 19, total * sizeof(Fr));
      Fr* s2 = (Fr*)ws(k.c, 20, total * sizeof(Fr));
      send = (Fr*)ws(k.c, 21, (size_t)n * out_per_party * sizeof(Fr));
      // transpose + unpack (dfft/mod.rs:207-220): s1[e*l + j] = sum_p U[j][p] * gathered[p][e]
      hipLaunchKernelGGL(matvec_kernel<Fr>, dim3(nblk(mbyl)), dim3(256), 0, k.s(), mat(pp, degree2 ? 2 : 1), l, n,
                         gathered, (size_t)1, mbyl, s1, (size_t)l, (size_t)1, mbyl, -1);
      // fft2_in_place (dfft/mod.rs:161-177)
      const TwiddleSet& ts = twiddles(k, curve, log_m, inverse);
      unsigned log_l = 0;
      while ((1u << log_l) < l) log_l++;
      Fr *a = s1, *b = s2;
      for (unsigned i = log_l; i >= 1; i--) {
        hipLaunchKernelGGL(fft2_level_kernel<Fr>, dim3(nblk(m / 2)), dim3(256), 0, k.s(), a, b, m, i, log_m,
                           (const Fr*)ts.lo, (const Fr*)ts.hi, ts.lb);
        std::swap(a, b);
      }
      hipLaunchKernelGGL(rotate_pad_kernel<Fr>, dim3(nblk(total)), dim3(256), 0, k.s(), a, b, m, total);
      // b now holds the padded natural-order vector
      if (rearrange) {
        // bit-reverse, then pack s1r[i], s1r[i + len/l], .. (dfft/mod.rs:230-245): in index = bitrev(i + j*len/l)
        unsigned bits = 0;
        while (((size_t)1 << bits) < total) bits++;
        hipLaunchKernelGGL(matvec_kernel<Fr>, dim3(nblk(out_per_party)), dim3(256), 0, k.s(), mat(pp, 0), n, l, b,
                           (size_t)1, out_per_party, send, (size_t)1, out_per_party, out_per_party, (int)bits);
      } else {
        // pack_vec over consecutive l-chunks (utils/pack.rs:4-16), transposed to [party][chunk]
        hipLaunchKernelGGL(matvec_kernel<Fr>, dim3(nblk(out_per_party)), dim3(256), 0, k.s(), mat(pp, 0), n, l, b,
                           (size_t)l, (size_t)1, send, (size_t)1, out_per_party, out_per_party, -1);
      }
      DG_HIP(hipGetLastError());
    }
    net_check(net->scatter_from_king(net->self, sid, send, out_per_party * sizeof(Fr), out, k.s()));
  }

  static const TwiddleSet& twiddles(Call& k, int curve, unsigned log_n, int inverse);

  // deg_red (utils/deg_red.rs:10-28) on device buffers
  static void deg_red(Call& k, const dg16_pss* pp, const dg16_net* net, const Fr* din, size_t count, Fr* dout, int sid) {
    const unsigned n = pp->n, l = pp->l;
    const bool king = net->party_id(net->self) == 0;
    Fr* gathered = king ? (Fr*)ws(k.c, 18, (size_t)n * count * sizeof(Fr)) : nullptr;
    net_check(net->gather_to_king(net->self, sid, din, count * sizeof(Fr), gathered, k.s()));
    Fr* send = nullptr;
    if (king) {
      Fr* sec = (Fr*)ws(k.c, 19, count * l * sizeof(Fr));
      send = (Fr*)ws(k.c, 21, (size_t)n * count * sizeof(Fr));
      // unpack2_in_place then pack_from_public_in_place per packed element
      hipLaunchKernelGGL(matvec_kernel<Fr>, dim3(nblk(count)), dim3(256), 0, k.s(), mat(pp, 2), l, n, gathered,
                         (size_t)1, count, sec, (size_t)l, (size_t)1, count, -1);
      hipLaunchKernelGGL(matvec_kernel<Fr>, dim3(nblk(count)), dim3(256), 0, k.s(), mat(pp, 0), n, l, sec, (size_t)l,
                         (size_t)1, send, (size_t)1, count, count, -1);
      DG_HIP(hipGetLastError());
    }
    net_check(net->scatter_from_king(net->self, sid, send, count * sizeof(Fr), dout, k.s()));
  }

  // d_fft / d_ifft (dfft/mod.rs:17-95)
  static void d_fft(Call& k, const dg16_pss* pp, const dg16_net* net, int curve, Fr* share, size_t mbyl, bool rearrange,
                    unsigned pad, bool degree2, unsigned log_m, int inverse, Fr* out, int sid) {
    const unsigned l = pp->l;
    DG_REQUIRE(mbyl * l == ((size_t)1 << log_m), DG16_ERR_BAD_ARG, "Mismatch of size in FFT (share.len() * l != dom.size())");
    const TwiddleSet& ts = twiddles(k, curve, log_m, inverse);
    if (inverse)   // peval_share *= dom.size_inv() (dfft/mod.rs:78)
      hipLaunchKernelGGL(scale_kernel<Fr>, dim3(nblk(mbyl)), dim3(256), 0, k.s(), share, (const Fr*)ts.n_inv, mbyl);
    unsigned log_l = 0;
    while ((1u << log_l) < l) log_l++;
    // fft1_in_place: levels i = log2 m down to log2 l + 1 (dfft/mod.rs:122-135)
    for (unsigned i = log_m; i > log_l; i--) {
      unsigned log_ps = log_m - i;
      hipLaunchKernelGGL(fft1_level_kernel<Fr>, dim3(nblk(mbyl / 2)), dim3(256), 0, k.s(), share, mbyl / 2, log_ps, i,
                         log_m, (const Fr*)ts.lo, (const Fr*)ts.hi, ts.lb);
    }
    DG_HIP(hipGetLastError());
    fft2_with_rearrange_pad(k, pp, net, curve, share, mbyl, rearrange, pad, degree2, log_m, inverse, out, sid);
  }
};

// twiddle sets are built by ntt.hip; re-declared here
const TwiddleSet& get_twiddles_any(Call& k, int curve, unsigned log_n, int inverse);
template <class Fr>
const TwiddleSet& Dist<Fr>::twiddles(Call& k, int curve, unsigned log_n, int inverse) {
  return get_twiddles_any(k, curve, log_n, inverse);
}

template <class Fr>
static void pss_build(dg16_pss* pp) {
  const unsigned n = pp->n, l = pp->l;
  DG_HIP(hipMalloc(&pp->mats, (6 * n * l + n) * sizeof(Fr)));
  hipLaunchKernelGGL(pss_setup_kernel<Fr>, dim3(1), dim3(256), 0, 0, l, (Fr*)pp->mats);
  DG_HIP(hipGetLastError());
  DG_HIP(hipDeviceSynchronize());
}

#define FR_SWITCH(curve, ...)                                   \
  switch (curve) {                                              \
    case 0: { using Fr = bn254_fr; __VA_ARGS__ } break;         \
    case 1: { using Fr = bls12_381_fr; __VA_ARGS__ } break;     \
    default: { using Fr = bls12_377_fr; __VA_ARGS__ } break;    \
  }

}  // namespace dg16

// per-(curve, group) entry points defined in msm_<curve>_g<k>.hip
namespace dg16 {
#define DECL_G(name)                                                                                         \
  void d_msm_##name(Call&, const dg16_pss*, const dg16_net*, int, const void*, const void*, size_t, unsigned, void*, \
                    const void*, unsigned, unsigned);                                                                       \
  void packexp_##name(Call&, const dg16_pss*, int, const void*, size_t, void*);                                   \
  void mpc_combine_##name(Call&, const void*, const void*, unsigned, unsigned, bool, void*);                      \
  void affine_to_jac_##name(Call&, const void*, void*);
DECL_G(bn254_g1) DECL_G(bn254_g2) DECL_G(bls12_381_g1) DECL_G(bls12_381_g2) DECL_G(bls12_377_g1) DECL_G(bls12_377_g2)
#define DISPATCH_G(fn, curve, group, ...)                                                        \
  switch ((curve) * 2 + (group) - 1) {                                                           \
    case 0: fn##_bn254_g1(__VA_ARGS__); break;                                                   \
    case 1: fn##_bn254_g2(__VA_ARGS__); break;                                                   \
    case 2: fn##_bls12_381_g1(__VA_ARGS__); break;                                               \
    case 3: fn##_bls12_381_g2(__VA_ARGS__); break;                                               \
    case 4: fn##_bls12_377_g1(__VA_ARGS__); break;                                               \
    case 5: fn##_bls12_377_g2(__VA_ARGS__); break;                                               \
    default: throw StatusError{DG16_ERR_BAD_ARG, "unknown (curve, group)"};                      \
  }
}  // namespace dg16

extern "C" {

// ---- PackedSharingParams ------------------------------------------------------------------------------
int dg16_pss_create(dg16_ctx* ctx, int curve, unsigned l, dg16_pss** out) {
  if (!ctx || !out) return DG16_ERR_BAD_ARG;
  *out = nullptr;
  dg16_pss* pp = new dg16_pss{ctx, curve, l, l - 1, 4 * l, nullptr};
  int rc = guarded(ctx, [&] {
    DG_REQUIRE(curve >= 0 && curve <= 2, DG16_ERR_BAD_CURVE, "unknown curve id");
    DG_REQUIRE(l >= 1 && l <= 8 && !(l & (l - 1)), DG16_ERR_BAD_ARG, "packing factor must be a power of two <= 8");
    DG_HIP(hipSetDevice(ctx->device));
    FR_SWITCH(curve, pss_build<Fr>(pp);)
  });
  if (rc != DG16_OK) { delete pp; return rc; }
  *out = pp;
  return DG16_OK;
}
void dg16_pss_destroy(dg16_pss* pp) {
  if (!pp) return;
  hipSetDevice(pp->ctx->device);
  if (pp->mats) hipFree(pp->mats);
  delete pp;
}

// batched pack_from_public / unpack / unpack2 (pss.rs:86-148): `count` independent packings.
// which: 0 pack ([count][l] -> [count][n]), 1 unpack ([count][n] -> [count][l]), 2 unpack2.
int dg16_pss_apply(dg16_ctx* ctx, const dg16_pss* pp, int which, const void* in, size_t count, void* out,
                   unsigned flags, int channel) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    DG_REQUIRE(pp && in && out && which >= 0 && which <= 2, DG16_ERR_BAD_ARG, "bad argument");
    const unsigned n = pp->n, l = pp->l;
    const unsigned cols = which == 0 ? l : n, rows = which == 0 ? n : l;
    bool dev = flags & DG16_F_DEVICE_PTRS;
    Call k(ctx, channel);
    FR_SWITCH(pp->curve, {
      const Fr* din = (const Fr*)stage_in(k, 0, in, count * cols * sizeof(Fr), dev);
      Fr* dout = dev ? (Fr*)out : (Fr*)ws(k.c, 1, count * rows * sizeof(Fr));
      if (count)
        hipLaunchKernelGGL(matvec_kernel<Fr>, dim3(nblk(count)), dim3(256), 0, k.s(), Dist<Fr>::mat(pp, which), rows,
                           cols, din, (size_t)cols, (size_t)1, dout, (size_t)rows, (size_t)1, count, -1);
      DG_HIP(hipGetLastError());
      if (!dev) stage_out(k, out, dout, count * rows * sizeof(Fr), false);
    })
    k.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k.s()));
  });
}

// ---- LocalNet -------------------------------------------------------------------------------------------
int dg16_localnet_create(unsigned n_parties, dg16_localnet** out) {
  if (!out || n_parties < 1 || n_parties > kMaxParties) return DG16_ERR_BAD_ARG;
  dg16_localnet* ln = new dg16_localnet();
  ln->n = n_parties;
  for (unsigned i = 0; i < n_parties; i++) {
    ln->ref[i] = {ln, i};
    ln->party[i] = dg16_net{&ln->ref[i], localnet_n, localnet_id, localnet_gather, localnet_scatter,
                            localnet_is_init, localnet_send_to, localnet_recv_from};
  }
  *out = ln;
  return DG16_OK;
}
const dg16_net* dg16_localnet_party(dg16_localnet* ln, unsigned id) {
  return (ln && id < ln->n) ? &ln->party[id] : nullptr;
}
void dg16_localnet_destroy(dg16_localnet* ln) { delete ln; }
// A party that fails outside a collective calls this so that its peers error out instead of waiting
// (the reference surfaces a dead peer as MpcNetError::Generic("Stream died"), mpc-net/src/multi.rs:393).
void dg16_localnet_abort(dg16_localnet* ln) {
  if (!ln) return;
  std::lock_guard<std::mutex> g(ln->mu);
  ln->aborted = true;
  ln->cv.notify_all();
}
void dg16_localnet_reset(dg16_localnet* ln, unsigned timeout_s) {
  if (!ln) return;
  std::lock_guard<std::mutex> g(ln->mu);
  ln->aborted = false;
  if (timeout_s) ln->timeout_s = timeout_s;
  for (auto& ch : ln->slot)
    for (auto& sl : ch) { sl.arrived = 0; sl.failed = false; }
  for (auto& ch : ln->box)
    for (auto& from : ch)
      for (auto& b : from) b = {};
}

// ---- d_fft / d_ifft -----------------------------------------------------------------------------------------
int dg16_d_fft(dg16_ctx* ctx, const dg16_pss* pp, const dg16_net* net, const void* share, size_t share_len,
               unsigned log_m, int rearrange, unsigned pad, int degree2, int inverse, void* out, unsigned flags,
               int channel) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    DG_REQUIRE(pp && net && share && out && pad >= 1, DG16_ERR_BAD_ARG, "bad argument");
    DG_REQUIRE(net->n_parties(net->self) == pp->n, DG16_ERR_BAD_ARG, "net.n_parties() != pp.n");
    // debug_assert_eq!(pcoeff_share.len() * pp.l, dom.size()) (dfft/mod.rs:31-37)
    DG_REQUIRE(share_len * pp->l == ((size_t)1 << log_m), DG16_ERR_BAD_ARG,
               "Mismatch of size in FFT (share.len() * l != dom.size())");
    const size_t mbyl = share_len, out_n = (size_t)pad * mbyl;
    bool dev = flags & DG16_F_DEVICE_PTRS;
    Call k(ctx, channel);
    FR_SWITCH(pp->curve, {
      Fr* work = (Fr*)ws(k.c, 0, mbyl * sizeof(Fr));   // the reference consumes its input Vec
      DG_HIP(hipMemcpyAsync(work, share, mbyl * sizeof(Fr), dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, k.s()));
      Fr* dout = dev ? (Fr*)out : (Fr*)ws(k.c, 1, out_n * sizeof(Fr));
      Dist<Fr>::d_fft(k, pp, net, pp->curve, work, mbyl, rearrange != 0, pad, degree2 != 0, log_m, inverse, dout, channel);
      if (!dev) stage_out(k, out, dout, out_n * sizeof(Fr), false);
    })
    k.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k.s()));
  });
}

// ---- d_msm (dmsm/mod.rs:70-98) and pack/unpack in the exponent (dmsm/mod.rs:7-68) -------------------------
int dg16_d_msm(dg16_ctx* ctx, const dg16_pss* pp, const dg16_net* net, int group, const void* bases,
               const void* scalars, size_t n_bases, size_t n_scalars, unsigned flags, int channel, void* out) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    DG_REQUIRE(pp && net && out && (group == 1 || group == 2), DG16_ERR_BAD_ARG, "bad argument");
    DG_REQUIRE(n_bases == n_scalars, DG16_ERR_LENGTH_MISMATCH,
               "bases and scalars differ in length (VariableBaseMSM::msm returns Err(min_len))");
    DG_REQUIRE(net->n_parties(net->self) == pp->n, DG16_ERR_BAD_ARG, "net.n_parties() != pp.n");
    bool dev = flags & DG16_F_DEVICE_PTRS;
    size_t pb = affine_bytes(pp->curve, group);
    Call k(ctx, channel);
    const void* dbases = stage_in(k, 0, bases, n_bases * pb, dev);
    const void* dscal = stage_in(k, 1, scalars, n_bases * 32, dev);
    void* dout = dev ? out : ws(k.c, 2, pb / 2 * 3);
    DISPATCH_G(d_msm, pp->curve, group, k, pp, net, channel, dbases, dscal, n_bases, msm_mode(flags), dout,
               nullptr, 0u, 1u)
    if (!dev) stage_out(k, out, dout, pb / 2 * 3, false);
    k.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k.s()));
  });
}

int dg16_d_msm_resident(dg16_ctx* ctx, const dg16_pss* pp, const dg16_net* net, const dg16_bases* bases,
                        const void* scalars, size_t n_scalars, unsigned flags, int channel, void* out) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    DG_REQUIRE(pp && net && out && bases && bases->ctx == ctx && bases->curve == pp->curve, DG16_ERR_BAD_ARG,
               "bad argument");
    DG_REQUIRE(bases->n == n_scalars, DG16_ERR_LENGTH_MISMATCH,
               "bases and scalars differ in length (VariableBaseMSM::msm returns Err(min_len))");
    DG_REQUIRE(net->n_parties(net->self) == pp->n, DG16_ERR_BAD_ARG, "net.n_parties() != pp.n");
    bool dev = flags & DG16_F_DEVICE_PTRS;
    size_t pb = affine_bytes(pp->curve, bases->group);
    Call k(ctx, channel);
    const void* dscal = stage_in(k, 1, scalars, n_scalars * 32, dev);
    void* dout = dev ? out : ws(k.c, 2, pb / 2 * 3);
    DISPATCH_G(d_msm, pp->curve, bases->group, k, pp, net, channel, nullptr, dscal, n_scalars,
               msm_mode(flags), dout, bases->table, bases->c, bases->stride)
    if (!dev) stage_out(k, out, dout, pb / 2 * 3, false);
    k.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k.s()));
  });
}

// which: 0 packexp_from_public ([count][l] -> [count][n]), 1 unpackexp degree t+l, 2 unpackexp degree2
int dg16_pss_apply_exp(dg16_ctx* ctx, const dg16_pss* pp, int group, int which, const void* in, size_t count,
                       void* out, unsigned flags, int channel) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    DG_REQUIRE(pp && in && out && which >= 0 && which <= 2 && (group == 1 || group == 2), DG16_ERR_BAD_ARG, "bad argument");
    const unsigned cols = which == 0 ? pp->l : pp->n, rows = which == 0 ? pp->n : pp->l;
    bool dev = flags & DG16_F_DEVICE_PTRS;
    size_t pb = affine_bytes(pp->curve, group);
    Call k(ctx, channel);
    const void* din = stage_in(k, 0, in, count * cols * pb, dev);
    void* dout = dev ? out : ws(k.c, 1, count * rows * pb);
    if (count) { DISPATCH_G(packexp, pp->curve, group, k, pp, which, din, count, dout) }
    if (!dev) stage_out(k, out, dout, count * rows * pb, false);
    k.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k.s()));
  });
}

// ---- deg_red (utils/deg_red.rs:10-28) ---------------------------------------------------------------------------
int dg16_deg_red(dg16_ctx* ctx, const dg16_pss* pp, const dg16_net* net, const void* px, size_t count, void* out,
                 unsigned flags, int channel) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    DG_REQUIRE(pp && net && px && out, DG16_ERR_BAD_ARG, "bad argument");
    bool dev = flags & DG16_F_DEVICE_PTRS;
    Call k(ctx, channel);
    FR_SWITCH(pp->curve, {
      const Fr* din = (const Fr*)stage_in(k, 0, px, count * sizeof(Fr), dev);
      Fr* dout = dev ? (Fr*)out : (Fr*)ws(k.c, 1, count * sizeof(Fr));
      Dist<Fr>::deg_red(k, pp, net, din, count, dout, channel);
      if (!dev) stage_out(k, out, dout, count * sizeof(Fr), false);
    })
    k.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k.s()));
  });
}

// ---- d_pp (dpp/mod.rs:17-88): packed shares of the prefix products of num[i]/den[i] ---------------------------------
int dg16_d_pp(dg16_ctx* ctx, const dg16_pss* pp, const dg16_net* net, const void* num, const void* den, size_t count,
              void* out, unsigned flags, int channel) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    DG_REQUIRE(pp && net && num && den && out, DG16_ERR_BAD_ARG, "bad argument");
    const unsigned n = pp->n, l = pp->l;
    bool dev = flags & DG16_F_DEVICE_PTRS;
    Call k(ctx, channel);
    FR_SWITCH(pp->curve, {
      // numden_rand = num*s ++ den*s with the dummy mask s = 1 (dpp/mod.rs:24-34)
      Fr* nd = (Fr*)ws(k.c, 0, 2 * count * sizeof(Fr));
      hipMemcpyKind kind = dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
      DG_HIP(hipMemcpyAsync(nd, num, count * sizeof(Fr), kind, k.s()));
      DG_HIP(hipMemcpyAsync(nd + count, den, count * sizeof(Fr), kind, k.s()));
      Fr* mid = (Fr*)ws(k.c, 2, count * sizeof(Fr));
      const bool king = net->party_id(net->self) == 0;
      Fr* gathered = king ? (Fr*)ws(k.c, 18, (size_t)n * 2 * count * sizeof(Fr)) : nullptr;
      net_check(net->gather_to_king(net->self, channel, nd, 2 * count * sizeof(Fr), gathered, k.s()));
      Fr* send = nullptr;
      if (king) {
        const size_t tot = 2 * count * l, half = count * l;
        Fr* numden = (Fr*)ws(k.c, 19, tot * sizeof(Fr));
        Fr* ratio = (Fr*)ws(k.c, 20, half * sizeof(Fr));
        send = (Fr*)ws(k.c, 21, (size_t)n * count * sizeof(Fr));
        // unpack2 of every packed element: numden = [num secrets (count*l) | den secrets (count*l)]
        hipLaunchKernelGGL(matvec_kernel<Fr>, dim3(nblk(2 * count)), dim3(256), 0, k.s(), Dist<Fr>::mat(pp, 2), l, n,
                           gathered, (size_t)1, 2 * count, numden, (size_t)l, (size_t)1, 2 * count, -1);
        hipLaunchKernelGGL(ratio_kernel<Fr>, dim3(nblk(half)), dim3(256), 0, k.s(), numden, ratio, half);   // :58-63
        size_t ntiles = (half + kScanTile - 1) / kScanTile;
        Fr* tops = (Fr*)ws(k.c, 22, ntiles * sizeof(Fr));
        hipLaunchKernelGGL(prefix_prod_tile_kernel<Fr>, dim3((unsigned)ntiles), dim3(256), 0, k.s(), ratio, half, tops);
        hipLaunchKernelGGL(prefix_prod_tops_kernel<Fr>, dim3(1), dim3(1), 0, k.s(), tops, ntiles);
        hipLaunchKernelGGL(prefix_prod_fix_kernel<Fr>, dim3(nblk(half)), dim3(256), 0, k.s(), ratio, half, tops);  // :66-69
        hipLaunchKernelGGL(matvec_kernel<Fr>, dim3(nblk(count)), dim3(256), 0, k.s(), Dist<Fr>::mat(pp, 0), n, l, ratio,
                           (size_t)l, (size_t)1, send, (size_t)1, count, count, -1);                                  // :74-79
        DG_HIP(hipGetLastError());
      }
      net_check(net->scatter_from_king(net->self, channel, send, count * sizeof(Fr), mid, k.s()));
      // sinv = 1: nothing to remove (dpp/mod.rs:86); then the degree reduction (dpp/mod.rs:87)
      Fr* dout = dev ? (Fr*)out : (Fr*)ws(k.c, 1, count * sizeof(Fr));
      Dist<Fr>::deg_red(k, pp, net, mid, count, dout, channel);
      if (!dev) stage_out(k, out, dout, count * sizeof(Fr), false);
    })
    k.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k.s()));
  });
}

// ---- ext_wit::h (groth16/src/ext_wit.rs:16-101) -------------------------------------------------------------------------
int dg16_ext_wit_h(dg16_ctx* ctx, const dg16_pss* pp, const dg16_net* net, const void* a_share, const void* b_share,
                   const void* c_share, unsigned log_m, void* out, unsigned flags) {
  if (!ctx) return DG16_ERR_BAD_ARG;
  return guarded(ctx, [&] {
    DG_REQUIRE(pp && net && a_share && b_share && c_share && out, DG16_ERR_BAD_ARG, "bad argument");
    const unsigned n = pp->n, l = pp->l;
    const size_t m = (size_t)1 << log_m, mbyl = m / l;
    bool dev = flags & DG16_F_DEVICE_PTRS;
    Call k(ctx, 0);
    FR_SWITCH(pp->curve, {
      const void* in[3] = {a_share, b_share, c_share};
      Fr* ev = (Fr*)ws(k.c, 2, 3 * 2 * mbyl * sizeof(Fr));      // three vectors of 2m/l evaluations
      Fr* work = (Fr*)ws(k.c, 0, 2 * mbyl * sizeof(Fr));
      Fr* coeff = (Fr*)ws(k.c, 1, 2 * mbyl * sizeof(Fr));
      for (int v = 0; v < 3; v++) {
        DG_HIP(hipMemcpyAsync(work, in[v], mbyl * sizeof(Fr), dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, k.s()));
        // d_ifft(.., rearrange = true, pad = 2) on the m domain, then d_fft on the 2m domain (ext_wit.rs:34-49)
        Dist<Fr>::d_fft(k, pp, net, pp->curve, work, mbyl, true, 2, false, log_m, 1, coeff, 0);
        DG_HIP(hipMemcpyAsync(work, coeff, 2 * mbyl * sizeof(Fr), hipMemcpyDeviceToDevice, k.s()));
        Dist<Fr>::d_fft(k, pp, net, pp->curve, work, 2 * mbyl, false, 1, false, log_m + 1, 0, ev + (size_t)v * 2 * mbyl, 0);
      }
      // king: unpack p, q, w, keep the odd positions, h = p*q - w, pack_vec, scatter (ext_wit.rs:54-96)
      const bool king = net->party_id(net->self) == 0;
      Fr* gathered = king ? (Fr*)ws(k.c, 18, (size_t)n * 3 * 2 * mbyl * sizeof(Fr)) : nullptr;
      net_check(net->gather_to_king(net->self, 0, ev, 3 * 2 * mbyl * sizeof(Fr), gathered, k.s()));
      Fr* send = nullptr;
      Fr* dout = dev ? (Fr*)out : (Fr*)ws(k.c, 3, mbyl * sizeof(Fr));
      if (king) {
        Fr* un = (Fr*)ws(k.c, 19, 3 * 2 * m * sizeof(Fr));
        Fr* h = (Fr*)ws(k.c, 20, m * sizeof(Fr));
        send = (Fr*)ws(k.c, 21, (size_t)n * mbyl * sizeof(Fr));
        for (int v = 0; v < 3; v++)   // party p's record holds its three vectors back to back
          hipLaunchKernelGGL(matvec_kernel<Fr>, dim3(nblk(2 * mbyl)), dim3(256), 0, k.s(), Dist<Fr>::mat(pp, 1), l, n,
                             gathered + (size_t)v * 2 * mbyl, (size_t)1, (size_t)3 * 2 * mbyl, un + (size_t)v * 2 * m,
                             (size_t)l, (size_t)1, 2 * mbyl, -1);
        hipLaunchKernelGGL(h_odd_kernel<Fr>, dim3(nblk(m)), dim3(256), 0, k.s(), un, un + 2 * m, un + 4 * m, h, m, l, pp->t);
        hipLaunchKernelGGL(matvec_kernel<Fr>, dim3(nblk(mbyl)), dim3(256), 0, k.s(), Dist<Fr>::mat(pp, 0), n, l, h,
                           (size_t)l, (size_t)1, send, (size_t)1, mbyl, mbyl, -1);
        DG_HIP(hipGetLastError());
      }
      net_check(net->scatter_from_king(net->self, 0, send, mbyl * sizeof(Fr), dout, k.s()));
      if (!dev) stage_out(k, out, dout, mbyl * sizeof(Fr), false);
    })
    k.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k.s()));
  });
}

}  // extern "C"

// ---- prove::A / B / C::compute (groth16/src/prove.rs:21-46, 62-85, 106-136) ------------------------------------------
namespace dg16 {
// one d_msm on `channel` (its stream, its workspace), Jacobian result at the device address `out_dev`
static void d_msm_on_channel(dg16_ctx* ctx, const dg16_pss* pp, const dg16_net* net, int group, const void* bases,
                             const void* scalars, size_t n_bases, size_t n_scalars, unsigned flags, int channel,
                             void* out_dev) {
  DG_REQUIRE(n_bases == n_scalars, DG16_ERR_LENGTH_MISMATCH,
             "bases and scalars differ in length (VariableBaseMSM::msm returns Err(min_len))");
  const bool dev = flags & DG16_F_DEVICE_PTRS;
  const size_t pb = affine_bytes(pp->curve, group);
  Call k(ctx, channel);
  const void* dbases = stage_in(k, 0, bases, n_bases * pb, dev);
  const void* dscal = stage_in(k, 1, scalars, n_bases * 32, dev);
  DISPATCH_G(d_msm, pp->curve, group, k, pp, net, channel, dbases, dscal, n_bases, msm_mode(flags), out_dev,
             nullptr, 0u, 1u)
  k.finish();
  DG_HIP(hipStreamSynchronize(k.s()));     // the caller combines on another channel's stream
}
// Term buffer of one compute call (channel-0 workspace slot 29): 8 Jacobian slots, then 8 scalars, then the result.
struct MpcTerms {
  uint8_t* base;
  size_t jb;
  void* term(unsigned i) const { return base + i * jb; }
  void* scalar(unsigned i) const { return base + 8 * jb + i * 32; }
  void* result() const { return base + 8 * jb + 8 * 32; }
};
// The buffer has ONE size for every curve and group (the largest Jacobian point is 6 x 48 bytes), so the grow-only slot
// is allocated once per context and never moves; ctx->mpc_mu serialises the prove_* calls of a context, which are the
// only users (a party runs A, B, C one after another: groth16/examples/sha256.rs:45-92).
constexpr size_t kMpcTermsBytes = 9 * 288 + 8 * 32;
static MpcTerms mpc_terms(Channel& c, int curve, int group) {
  const size_t jb = affine_bytes(curve, group) / 2 * 3;
  static_assert(kMpcTermsBytes >= 9 * 288 + 8 * 32, "term buffer");
  return MpcTerms{(uint8_t*)ws(c, 29 + (group - 1), kMpcTermsBytes), jb};
}
// affine point handed over by the caller (host or device) -> Jacobian term slot
static void put_affine_term(Call& k, int curve, int group, const void* pt, bool dev, int stage_slot, void* term) {
  const void* d = stage_in(k, stage_slot, pt, affine_bytes(curve, group), dev);
  DISPATCH_G(affine_to_jac, curve, group, k, d, term)
}
static void put_scalar(Call& k, const void* sc, bool dev, void* dst) {
  DG_HIP(hipMemcpyAsync(dst, sc, 32, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, k.s()));
}
// A = L + N * r + d_msm(S, a) in G1 (prove.rs:36-44)  /  B = Z + K * s + d_msm(V, a) in G2 (prove.rs:76-83)
static void prove_ab(dg16_ctx* ctx, const dg16_pss* pp, const dg16_net* net, int group, const void* fixed0,
                     const void* fixed1, const void* k1, const void* bases, const void* scalars, size_t n_bases,
                     size_t n_scalars, unsigned flags, int channel, void* out) {
  DG_REQUIRE(pp && net && fixed0 && fixed1 && k1 && out, DG16_ERR_BAD_ARG, "bad argument");
  DG_REQUIRE(net->n_parties(net->self) == pp->n, DG16_ERR_BAD_ARG, "net.n_parties() != pp.n");
  const bool dev = flags & DG16_F_DEVICE_PTRS;
  std::lock_guard<std::mutex> whole_call(ctx->mpc_mu);
  MpcTerms t;
  {
    Call k(ctx, channel);
    t = mpc_terms(k.c, pp->curve, group);
  }
  d_msm_on_channel(ctx, pp, net, group, bases, scalars, n_bases, n_scalars, flags, channel, t.term(2));
  Call k(ctx, channel);
  put_affine_term(k, pp->curve, group, fixed0, dev, 2, t.term(0));      // L / Z
  put_affine_term(k, pp->curve, group, fixed1, dev, 3, t.term(1));      // N / K
  put_scalar(k, k1, dev, t.scalar(1));                                  // r / s
  DISPATCH_G(mpc_combine, pp->curve, group, k, t.term(0), t.scalar(0), 3u, 0x2u, flags & DG16_F_SCALARS_MONT, t.result())
  stage_out(k, out, t.result(), t.jb, dev);
  k.finish();
  if (!dev) DG_HIP(hipStreamSynchronize(k.s()));
}
}  // namespace dg16

extern "C" {

int dg16_prove_a(dg16_ctx* ctx, const dg16_pss* pp, const dg16_net* net, const void* L, const void* N, const void* r,
                 const void* S, const void* a, size_t n_S, size_t n_a, unsigned flags, int channel, void* out) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] { prove_ab(ctx, pp, net, 1, L, N, r, S, a, n_S, n_a, flags, channel, out); });
}

int dg16_prove_b(dg16_ctx* ctx, const dg16_pss* pp, const dg16_net* net, const void* Z, const void* K, const void* s,
                 const void* V, const void* a, size_t n_V, size_t n_a, unsigned flags, int channel, void* out) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] { prove_ab(ctx, pp, net, 2, Z, K, s, V, a, n_V, n_a, flags, channel, out); });
}

// C = w + u + A * s + M * r + h * r with w = d_msm(W, ax), u = d_msm(U, h), h = d_msm(H, a) JOINED on channels 0 / 1 / 2
// (prove.rs:113-125: three futures under tokio::try_join!): three host threads drive the three channels of this one
// context at the same time -- each channel has its own stream, workspace and lock (ctx.h), and the transport keeps one
// rendezvous slot per channel -- then channel 0 combines (prove.rs:127-134).
int dg16_prove_c(dg16_ctx* ctx, const dg16_pss* pp, const dg16_net* net, const void* A, const void* M, const void* s,
                 const void* r, const void* W, const void* ax, size_t n_W, size_t n_ax, const void* U, const void* h,
                 size_t n_U, size_t n_h, const void* H, const void* a, size_t n_H, size_t n_a, unsigned flags,
                 void* out) {
  if (!ctx) return DG16_ERR_BAD_ARG;
  return guarded(ctx, [&] {
    DG_REQUIRE(pp && net && A && M && s && r && out, DG16_ERR_BAD_ARG, "bad argument");
    DG_REQUIRE(net->n_parties(net->self) == pp->n, DG16_ERR_BAD_ARG, "net.n_parties() != pp.n");
    const bool dev = flags & DG16_F_DEVICE_PTRS;
    std::lock_guard<std::mutex> whole_call(ctx->mpc_mu);
    MpcTerms t;
    {
      Call k(ctx, 0);
      t = mpc_terms(k.c, pp->curve, 1);
    }
    // terms: 0 w, 1 u, 2 A (* s), 3 M (* r), 4 h (* r)
    struct Job { const void *bases, *scalars; size_t nb, ns; unsigned slot; } jobs[3] = {
        {W, ax, n_W, n_ax, 0}, {U, h, n_U, n_h, 1}, {H, a, n_H, n_a, 4}};
    StatusError errs[3] = {{DG16_OK, ""}, {DG16_OK, ""}, {DG16_OK, ""}};
    auto run = [&](int c) {
      try {
        d_msm_on_channel(ctx, pp, net, 1, jobs[c].bases, jobs[c].scalars, jobs[c].nb, jobs[c].ns, flags, c,
                         t.term(jobs[c].slot));
      } catch (const StatusError& e) {
        errs[c] = e;
      } catch (const std::exception& e) {
        errs[c] = StatusError{DG16_ERR_HIP, e.what()};
      }
    };
    // The library's own RCCL net runs the channels IN ORDER unless DG16_RCCL_JOIN_CHANNELS=1: three communicators driven
    // concurrently from three host threads in an order that differs per rank is NCCL's documented deadlock hazard (all
    // their kernels must be co-resident), each d_msm saturates the device anyway, and no multi-GPU box has ever run the
    // joined form -- the safe form is the default there, the joined one an opt-in.
    static const bool rccl_join = [] { const char* e = getenv("DG16_RCCL_JOIN_CHANNELS"); return e && e[0] == '1'; }();
    if ((flags & DG16_F_SERIAL_CHANNELS) || (net_is_rccl(net) && !rccl_join)) {
      // a transport whose channels are NOT independent (one ordered pipe for all of them): the three d_msm one after
      // another in the fixed order 0, 1, 2 on every party -- same result, no concurrency
      for (int c = 0; c < 3 && (c == 0 || errs[c - 1].code == DG16_OK); c++) run(c);
    } else {
      std::thread th[3];
      for (int c = 0; c < 3; c++) th[c] = std::thread(run, c);
      for (auto& x : th) x.join();
    }
    for (const auto& e : errs)
      if (e.code != DG16_OK) throw e;
    Call k(ctx, 0);
    const size_t jb = t.jb;
    DG_HIP(hipMemcpyAsync(t.term(2), A, jb, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, k.s()));   // A: E::G1
    put_affine_term(k, pp->curve, 1, M, dev, 2, t.term(3));
    put_scalar(k, s, dev, t.scalar(2));
    put_scalar(k, r, dev, t.scalar(3));
    put_scalar(k, r, dev, t.scalar(4));
    DISPATCH_G(mpc_combine, pp->curve, 1, k, t.term(0), t.scalar(0), 5u, 0x1Cu, flags & DG16_F_SCALARS_MONT, t.result())
    stage_out(k, out, t.result(), jb, dev);
    k.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k.s()));
  });
}

}  // extern "C"
