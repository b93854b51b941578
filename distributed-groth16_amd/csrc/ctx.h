// Host-side context of libdg16: one per process and GPU, three channels (HIP stream + grow-only
// device workspace each), mirroring MultiplexedStreamID::{Zero,One,Two} of the reference
// (mpc-net/src/lib.rs:29-33).  Thread-safe per channel (callers are concurrent host threads, like
// the reference's tokio tasks: mpc-net/src/multi.rs:305-314).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dg16.h"

namespace dg16 {

constexpr int kChannels = 3;
constexpr int kSlots = 32;

struct Channel {
  hipStream_t own = nullptr;
  hipStream_t cur = nullptr;
  std::mutex mu;
  void* slot[kSlots] = {};
  size_t slot_bytes[kSlots] = {};
  hipEvent_t ev[4] = {};
  bool ev_valid[2] = {false, false};  // [0] whole call (ev0..ev1), [1] dominant kernel (ev2..ev3)
};

struct TwiddleKey {
  int curve;
  unsigned log_n;
  int inverse;
  bool operator<(const TwiddleKey& o) const {
    if (curve != o.curve) return curve < o.curve;
    if (log_n != o.log_n) return log_n < o.log_n;
    return inverse < o.inverse;
  }
};
struct TwiddleSet {   // all device pointers, 32-byte Fr elements
  void* lo = nullptr;        // w^j, j < 2^lb
  void* hi = nullptr;        // w^(j << lb), j < 2^(log_n - lb)
  void* hi_scaled = nullptr; // hi * n^-1 (inverse only)
  void* small = nullptr;     // w_{2^sm}^t, t < 2^(sm-1)
  void* n_inv = nullptr;     // one element
  unsigned lb = 0, sm = 0;
  // the same tables in the reduced-radix Montgomery form of fp29.h (x R, R = 2^261; packed, canonical): what the
  // NTT kernels multiply by -- the arkworks-form tables above serve the dist-primitives mirror (dist_impl.h)
  void *lo_i = nullptr, *hi_i = nullptr, *hi_scaled_i = nullptr, *small_i = nullptr, *n_inv_i = nullptr;
  // the inter-step twiddles of the first / second step of a multi-step plan as ONE table each, in the step's data layout
  // (ntt.hip: full_twiddle_kernel), and w^o for o < n / 2 flattened (the w_2m shift of the h-polynomial); null = not built
  void* full[2] = {nullptr, nullptr};
  void* shift_full = nullptr;
  bool shift_full_tried = false;   // the optional table's allocation is attempted once per cache entry
};

}  // namespace dg16

struct dg16_ctx {
  int device = 0;
  int compute_units = 0;
  std::string name;
  dg16::Channel ch[dg16::kChannels];
  std::mutex mu;  // guards err + twiddle cache
  std::mutex mpc_mu;  // one prove::A / B / C::compute at a time per context (they share the term buffers of dist.hip)
  std::string err;
  // Long-lived events for cross-stream dependencies inside one call (prover pipeline).  They are never
  // destroyed while the context lives: destroying an event right after hipStreamWaitEvent() let later
  // launches overtake the wait once all buffers were warm (second proof on a context differed from the
  // oracle; the first one was masked by the implicit synchronisation of hipMalloc).
  hipEvent_t pipe_ev[24] = {};
  // DG16_F_OVERLAP_TAIL (prover_impl.h: prove_typed): the last proof's H reduction + assembly + copy-out may still be
  // running on channel 2's stream when the call returns; pipe_ev[18] is recorded behind them.  Every later call on this
  // context waits for it on its own stream (Call) -- except the calls that are known not to touch what the tail uses
  // (R1CS x witness on device pointers; the next overlapped proof, which places the waits itself).
  std::atomic<bool> tail_pending{false};
  hipStream_t aux[1] = {};   // extra internal stream of the prover pipeline (never handed out): B1's reduction, the exchanges
  dg16::Channel xws[2];      // workspace-only (no stream): the bucket buffers of the H and L MSMs of a proof
  // Sticky argument-error flag of stream-ordered calls (pinned host word mapped into the device: kernels OR
  // bits into it, dg16_sync reads it after the stream has drained).  bit 0: dg16_qap index out of range.
  unsigned* dev_flag_host = nullptr;
  unsigned* dev_flag = nullptr;
  // ClkProbe counters of the accumulation kernels (msm_impl.h): two device words per channel -- shader ticks, 100-MHz ticks
  // of the most recent accumulation whose timing events live on that channel (dg16_last_kernel_ms, which = 2)
  unsigned long long* kclk = nullptr;
  // HBM budget of the window tables of ONE resident key / base set built from here on (dg16_ctx_set_table_budget);
  // 0 = unlimited (one table row per window)
  size_t table_budget = 0;
  std::map<dg16::TwiddleKey, dg16::TwiddleSet> twiddles;
};

struct dg16_bases {          // resident bases: table of window multiples (dg16_bases_upload)
  dg16_ctx* ctx = nullptr;
  int curve = 0, group = 1;
  size_t n = 0;
  unsigned c = 0, nwin = 0;     // window bits; table rows
  unsigned stride = 1;          // the table keeps every stride-th window's row (1 = all)
  void* table = nullptr;
  size_t bytes = 0;
};

namespace dg16 {

struct StatusError {
  int code;
  std::string msg;
};

#define DG_HIP(expr)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      throw dg16::StatusError{e_ == hipErrorOutOfMemory ? DG16_ERR_OOM : DG16_ERR_HIP,      \
                              std::string(#expr) + ": " + hipGetErrorString(e_)};           \
  } while (0)

#define DG_REQUIRE(cond, code, text)                                                        \
  do {                                                                                      \
    if (!(cond)) throw dg16::StatusError{code, text};                                       \
  } while (0)

// grow-only workspace slot; contents are undefined after a grow
inline void* ws(Channel& c, int slot, size_t bytes) {
  if (bytes == 0) bytes = 16;
  if (c.slot_bytes[slot] < bytes) {
    if (c.slot[slot]) {
      DG_HIP(hipDeviceSynchronize());   // other streams may still read the old buffer (prover pipelines)
      DG_HIP(hipFree(c.slot[slot]));
      c.slot[slot] = nullptr;
      c.slot_bytes[slot] = 0;
    }
    size_t want = bytes + bytes / 8;
    DG_HIP(hipMalloc(&c.slot[slot], want));
    c.slot_bytes[slot] = want;
  }
  return c.slot[slot];
}

// RAII: lock the channel, make the device current, record whole-call events
struct Call {
  dg16_ctx* ctx;
  Channel& c;
  std::unique_lock<std::mutex> lk;
  Call(dg16_ctx* ctx_, int channel, bool tail_ok = false)
      : ctx(ctx_), c(ctx_->ch[channel]), lk(ctx_->ch[channel].mu) {
    DG_HIP(hipSetDevice(ctx->device));
    if (!tail_ok && ctx->tail_pending.load(std::memory_order_acquire))
      DG_HIP(hipStreamWaitEvent(c.cur, ctx->pipe_ev[18], 0));      // an overlapped proof's tail (see dg16_ctx::tail_pending)
    c.ev_valid[0] = c.ev_valid[1] = false;
    DG_HIP(hipEventRecord(c.ev[0], c.cur));
  }
  void begin_dominant() { DG_HIP(hipEventRecord(c.ev[2], c.cur)); }
  void end_dominant() {
    DG_HIP(hipEventRecord(c.ev[3], c.cur));
    c.ev_valid[1] = true;
  }
  void finish() {
    DG_HIP(hipEventRecord(c.ev[1], c.cur));
    c.ev_valid[0] = true;
  }
  hipStream_t s() const { return c.cur; }
};

inline int guard_channel(dg16_ctx* ctx, int channel) {
  if (!ctx) return DG16_ERR_BAD_ARG;
  if (channel < 0 || channel >= kChannels) {
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->err = "channel must be 0..2";
    return DG16_ERR_BAD_ARG;
  }
  return DG16_OK;
}

template <class Fn>
int guarded(dg16_ctx* ctx, Fn&& fn) {
  try {
    fn();
    return DG16_OK;
  } catch (const StatusError& e) {
    if (ctx) {
      std::lock_guard<std::mutex> g(ctx->mu);
      ctx->err = e.msg;
    }
    return e.code;
  } catch (const std::exception& e) {
    if (ctx) {
      std::lock_guard<std::mutex> g(ctx->mu);
      ctx->err = e.what();
    }
    return DG16_ERR_HIP;
  } catch (...) {
    return DG16_ERR_HIP;
  }
}

// staging helpers: host-pointer calls copy through workspace slots
inline const void* stage_in(Call& k, int slot, const void* p, size_t bytes, bool device_ptrs) {
  if (device_ptrs || bytes == 0) return p;
  void* d = ws(k.c, slot, bytes);
  DG_HIP(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, k.s()));
  return d;
}
inline void stage_out(Call& k, void* host_dst, const void* dev_src, size_t bytes, bool device_ptrs) {
  if (device_ptrs) {
    if (host_dst != dev_src) DG_HIP(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToDevice, k.s()));
  } else {
    DG_HIP(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, k.s()));
  }
}

// implemented in the .hip translation units
void field_op_launch(Call& k, int field_id, int op, const void* a, const void* b, void* out, size_t n);
void qap_launch(Call& k, int curve, const unsigned* a_ptr, const unsigned* a_col, const void* a_val,
                const unsigned* b_ptr, const unsigned* b_col, const void* b_val, const void* w, bool w_mont, size_t nc,
                size_t ni, size_t nv, size_t m, size_t row_start, size_t row_stride, void* a, void* b, void* c);
void h_poly_dist_stage(Call& k, int curve, unsigned log_m, unsigned rank, unsigned n_ranks, int stage,
                       const void* const* in, void* out);
void h_poly_dist_launch(Call& k, int curve, const dg16_comm* comm, const void* a, const void* b, const void* c,
                        unsigned log_m, void* out);
void ntt_dist_stage(Call& k, int curve, unsigned log_n_total, unsigned rank, unsigned n_ranks, int inverse, int stage,
                    const void* in, void* out);
void ntt_dist_launch(Call& k, int curve, const dg16_comm* comm, const void* in, void* out, unsigned log_n_total,
                     int inverse);
void ntt_launch(Call& k, int curve, void* data, unsigned log_n, int inverse, const void* coset_host);
void h_poly_launch(Call& k, int curve, const void* a, const void* b, const void* c, unsigned log_m,
                   void* out);
bool net_is_rccl(const dg16_net* net);     // rccl_net.hip: the library's own RCCL transport
// `mode` of the plain-MSM entry points: bit 0 = scalars in Montgomery form, bit 1 = DG16_F_BASES_IN_SUBGROUP
inline unsigned msm_mode(unsigned flags) {
  return ((flags & DG16_F_SCALARS_MONT) ? 1u : 0u) | ((flags & DG16_F_BASES_IN_SUBGROUP) ? 2u : 0u);
}
void msm_launch(Call& k, int curve, int group, const void* bases, const void* scalars, size_t n,
                unsigned mode, bool out_affine, void* out_dev);
void gen_bases_launch(Call& k, int curve, int group, uint64_t seed, size_t n, void* out_dev);
void to_affine_launch(Call& k, int curve, int group, const void* jac, void* out, size_t n);
void* bases_table_launch(Call& k, int curve, int group, const void* bases, size_t n, size_t budget, unsigned* c,
                         unsigned* rows, unsigned* stride);
void msm_resident_launch(Call& k, int curve, int group, const void* table, size_t n, unsigned c, unsigned stride,
                         const void* scalars, bool mont, bool affine, void* out);
size_t fq_bytes(int curve);
size_t affine_bytes(int curve, int group);

}  // namespace dg16
