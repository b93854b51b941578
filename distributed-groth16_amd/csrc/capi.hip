// C ABI of libdg16 (include/dg16.h): argument checking, host<->device staging, error mapping.
// All compute is in the HIP translation units next to this file; there is no CPU path.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bounds.h"
#include "ctx.h"

using namespace dg16;

namespace dg16 {
size_t fq_bytes(int curve) { return curve == DG16_BN254 ? 32 : 48; }
size_t affine_bytes(int curve, int group) { return 2 * fq_bytes(curve) * (group == 2 ? 2 : 1); }
#ifdef DG16_BOUNDS
// the violation record of the bounds-checked build (bounds.h): one zeroed device buffer per process
unsigned* bounds_sink_device() {
  static unsigned* sink = [] {
    unsigned* p = nullptr;
    if (hipMalloc((void**)&p, kBoundsWords * sizeof(unsigned)) != hipSuccess) return (unsigned*)nullptr;
    (void)hipMemset(p, 0, kBoundsWords * sizeof(unsigned));
    return p;
  }();
  return sink;
}
#endif
static void check_curve_group(int curve, int group) {
  DG_REQUIRE(curve >= 0 && curve <= 2, DG16_ERR_BAD_CURVE, "unknown curve id");
  DG_REQUIRE(group == 1 || group == 2, DG16_ERR_BAD_ARG, "group must be 1 (G1) or 2 (G2)");
}
}  // namespace dg16

extern "C" {

int dg16_ctx_create(int device, dg16_ctx** out) {
  if (!out) return DG16_ERR_BAD_ARG;
  *out = nullptr;
  dg16_ctx* ctx = new dg16_ctx();
  int rc = guarded(ctx, [&] {
    int count = 0;
    DG_HIP(hipGetDeviceCount(&count));
    DG_REQUIRE(count > 0, DG16_ERR_HIP, "no HIP device visible: libdg16 has no CPU path");
    DG_REQUIRE(device >= 0 && device < count, DG16_ERR_BAD_ARG, "device index out of range");
    DG_HIP(hipSetDevice(device));
    hipDeviceProp_t p;
    DG_HIP(hipGetDeviceProperties(&p, device));
    ctx->device = device;
    ctx->compute_units = p.multiProcessorCount;
    ctx->name = p.gcnArchName;
    // Channel 0 carries the saturating kernels of a proof; channels 1, 2 and the aux streams carry the latency-bound
    // bucket reductions that hide behind them: those get the higher priority, so that their few waves are scheduled
    // ahead of the thousands of queued accumulation waves.  (Measured and settled in round 4,
    // profiles/r4prio_side_priority_cu_reserve_ab.txt: no priority, or channel 0 confined to all but k CUs through
    // hipExtStreamCreateWithCUMask, changed nothing under the queue schedule -- the switches are gone.)
    int prio_lo = 0, prio_hi = 0;
    DG_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    for (int i = 0; i < kChannels; i++) {
      DG_HIP(hipStreamCreateWithPriority(&ctx->ch[i].own, hipStreamNonBlocking, i > 0 ? prio_hi : prio_lo));
      ctx->ch[i].cur = ctx->ch[i].own;
      for (int e = 0; e < 4; e++) DG_HIP(hipEventCreate(&ctx->ch[i].ev[e]));
    }
    for (auto& e : ctx->pipe_ev) DG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    DG_HIP(hipStreamCreateWithPriority(&ctx->aux[0], hipStreamNonBlocking, prio_hi));
    DG_HIP(hipHostMalloc((void**)&ctx->dev_flag_host, sizeof(unsigned), hipHostMallocMapped));
    *ctx->dev_flag_host = 0;
    DG_HIP(hipHostGetDevicePointer((void**)&ctx->dev_flag, ctx->dev_flag_host, 0));
    DG_HIP(hipMalloc((void**)&ctx->kclk, kChannels * 2 * sizeof(unsigned long long)));
    DG_HIP(hipMemset(ctx->kclk, 0, kChannels * 2 * sizeof(unsigned long long)));
  });
  if (rc != DG16_OK) {
    // keep the message reachable for the caller that failed to get a context
    static thread_local std::string last;
    last = ctx->err;
    delete ctx;
    return rc;
  }
  *out = ctx;
  return DG16_OK;
}

void dg16_ctx_destroy(dg16_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  hipDeviceSynchronize();
  for (int i = 0; i < kChannels; i++) {
    for (int s = 0; s < kSlots; s++)
      if (ctx->ch[i].slot[s]) hipFree(ctx->ch[i].slot[s]);
    for (int e = 0; e < 4; e++)
      if (ctx->ch[i].ev[e]) hipEventDestroy(ctx->ch[i].ev[e]);
    if (ctx->ch[i].own) hipStreamDestroy(ctx->ch[i].own);
  }
  for (auto& x : ctx->xws)
    for (int s = 0; s < kSlots; s++)
      if (x.slot[s]) hipFree(x.slot[s]);
  for (auto& e : ctx->pipe_ev)
    if (e) hipEventDestroy(e);
  for (auto& st : ctx->aux)
    if (st) hipStreamDestroy(st);
  if (ctx->dev_flag_host) hipHostFree(ctx->dev_flag_host);
  if (ctx->kclk) hipFree(ctx->kclk);
  for (auto& kv : ctx->twiddles) {
    hipFree(kv.second.lo);
    hipFree(kv.second.hi);
    if (kv.second.hi_scaled) hipFree(kv.second.hi_scaled);
    hipFree(kv.second.small);
    hipFree(kv.second.n_inv);
    for (void* q : {kv.second.lo_i, kv.second.hi_i, kv.second.hi_scaled_i, kv.second.small_i, kv.second.n_inv_i,
                    kv.second.full[0], kv.second.full[1], kv.second.shift_full})
      if (q) hipFree(q);
  }
  delete ctx;
}

const char* dg16_last_error(dg16_ctx* ctx) {
  if (!ctx) return "null context";
  std::lock_guard<std::mutex> g(ctx->mu);
  static thread_local std::string copy;
  copy = ctx->err;
  return copy.c_str();
}

int dg16_set_stream(dg16_ctx* ctx, int channel, void* hip_stream) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  std::lock_guard<std::mutex> g(ctx->ch[channel].mu);
  ctx->ch[channel].cur = hip_stream ? (hipStream_t)hip_stream : ctx->ch[channel].own;
  return DG16_OK;
}

int dg16_sync(dg16_ctx* ctx, int channel) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    DG_HIP(hipSetDevice(ctx->device));
    DG_HIP(hipStreamSynchronize(ctx->ch[channel].cur));
    // sticky device-side argument errors of stream-ordered (device-pointer) calls surface here
    if (ctx->dev_flag_host && *ctx->dev_flag_host) {
      const unsigned f = *ctx->dev_flag_host;
      *ctx->dev_flag_host = 0;
      throw StatusError{DG16_ERR_BAD_ARG, f & 1 ? "dg16_qap: coefficient out of range (column >= num_vars or bad row_ptr)"
                                                : "device-side argument check failed"};
    }
#ifdef DG16_BOUNDS
    if (unsigned* sink = bounds_sink_device()) {
      unsigned rec[kBoundsWords] = {};
      DG_HIP(hipDeviceSynchronize());
      DG_HIP(hipMemcpy(rec, sink, sizeof rec, hipMemcpyDeviceToHost));
      if (rec[0]) {
        DG_HIP(hipMemset(sink, 0, sizeof rec));
        char msg[256];
        snprintf(msg, sizeof msg, "DG16_BOUNDS: %u violation(s); first: site %u index %llu limit %llu (workgroup %u, lane %u)", rec[0],
                 rec[1], (unsigned long long)(rec[2] | ((uint64_t)rec[3] << 32)),
                 (unsigned long long)(rec[4] | ((uint64_t)rec[5] << 32)), rec[6], rec[7]);
        fprintf(stderr, "[dg16] %s\n", msg);
        throw StatusError{DG16_ERR_HIP, msg};
      }
    }
#endif
  });
}

int dg16_device_info(dg16_ctx* ctx, char* name, size_t name_len, int* compute_units) {
  if (!ctx) return DG16_ERR_BAD_ARG;
  if (name && name_len) {
    strncpy(name, ctx->name.c_str(), name_len - 1);
    name[name_len - 1] = 0;
  }
  if (compute_units) *compute_units = ctx->compute_units;
  return DG16_OK;
}

int dg16_last_kernel_ms(dg16_ctx* ctx, int channel, int which, float* ms) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  if (!ms || which < 0 || which > 2) return DG16_ERR_BAD_ARG;
  return guarded(ctx, [&] {
    Channel& c = ctx->ch[channel];
    std::lock_guard<std::mutex> g(c.mu);
    *ms = 0.f;
    if (which == 2) {      // the shader clock (MHz) the chip held under that accumulation kernel, measured by the kernel
      if (!c.ev_valid[1] || !ctx->kclk) return;
      DG_HIP(hipSetDevice(ctx->device));
      DG_HIP(hipEventSynchronize(c.ev[3]));
      unsigned long long t[2] = {0, 0};
      DG_HIP(hipMemcpy(t, ctx->kclk + 2 * channel, sizeof t, hipMemcpyDeviceToHost));
      if (t[1]) *ms = (float)((double)t[0] / (double)t[1] * 100.0);
      return;
    }
    if (!c.ev_valid[which]) return;
    DG_HIP(hipSetDevice(ctx->device));
    hipEvent_t a = c.ev[which == 0 ? 0 : 2], b = c.ev[which == 0 ? 1 : 3];
    DG_HIP(hipEventSynchronize(b));
    DG_HIP(hipEventElapsedTime(ms, a, b));
  });
}

int dg16_field_op(dg16_ctx* ctx, int field_id, int op, const void* a, const void* b, void* out,
                  size_t n, unsigned flags, int channel) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    int curve = field_id & 15;
    DG_REQUIRE((field_id & ~16) >= 0 && curve <= 2 && (field_id >> 4) <= 1, DG16_ERR_BAD_CURVE,
               "field id must be curve (Fq) or 16 + curve (Fr)");
    DG_REQUIRE(op >= 0 && op <= 7, DG16_ERR_BAD_ARG, "unknown field op");
    DG_REQUIRE(a && out && (b || op >= 3), DG16_ERR_BAD_ARG, "null operand");
    bool dev = flags & DG16_F_DEVICE_PTRS;
    size_t eb = (field_id & 16) ? 32 : fq_bytes(curve);
    Call k(ctx, channel);
    const void* da = stage_in(k, 0, a, n * eb, dev);
    const void* db = b ? stage_in(k, 1, b, n * eb, dev) : da;
    void* dout = dev ? out : ws(k.c, 2, n * eb);
    if (n) field_op_launch(k, field_id, op, da, db, dout, n);
    if (!dev) stage_out(k, out, dout, n * eb, false);
    k.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k.s()));
  });
}

int dg16_ntt(dg16_ctx* ctx, int curve, void* data, unsigned log_n, int inverse,
             const void* coset_offset, unsigned flags, int channel) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    DG_REQUIRE(curve >= 0 && curve <= 2, DG16_ERR_BAD_CURVE, "unknown curve id");
    DG_REQUIRE(data, DG16_ERR_BAD_ARG, "null data");
    const unsigned two_adicity[3] = {28, 32, 47};
    DG_REQUIRE(log_n <= two_adicity[curve] && log_n <= 30, DG16_ERR_BAD_ARG,
               "domain larger than the field's 2-adic subgroup (PolynomialDegreeTooLarge)");
    bool dev = flags & DG16_F_DEVICE_PTRS;
    size_t bytes = ((size_t)32) << log_n;
    Call k(ctx, channel);
    void* d = dev ? data : ws(k.c, 0, bytes);
    if (!dev) DG_HIP(hipMemcpyAsync(d, data, bytes, hipMemcpyHostToDevice, k.s()));
    ntt_launch(k, curve, d, log_n, inverse, coset_offset);
    if (!dev) stage_out(k, data, d, bytes, false);
    k.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k.s()));
  });
}

int dg16_h_poly(dg16_ctx* ctx, int curve, const void* a, const void* b, const void* c,
                unsigned log_m, void* out, unsigned flags, int channel) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    DG_REQUIRE(curve >= 0 && curve <= 2, DG16_ERR_BAD_CURVE, "unknown curve id");
    DG_REQUIRE(a && b && c && out, DG16_ERR_BAD_ARG, "null operand");
    const unsigned two_adicity[3] = {28, 32, 47};
    DG_REQUIRE(log_m + 1 <= two_adicity[curve] && log_m <= 29, DG16_ERR_BAD_ARG,
               "domain larger than the field's 2-adic subgroup (PolynomialDegreeTooLarge)");
    bool dev = flags & DG16_F_DEVICE_PTRS;
    size_t bytes = ((size_t)32) << log_m;
    Call k(ctx, channel);
    const void* da = stage_in(k, 0, a, bytes, dev);
    const void* db = stage_in(k, 1, b, bytes, dev);
    const void* dc = stage_in(k, 2, c, bytes, dev);
    void* dout = dev ? out : ws(k.c, 3, bytes);
    h_poly_launch(k, curve, da, db, dc, log_m, dout);
    if (!dev) stage_out(k, out, dout, bytes, false);
    k.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k.s()));
  });
}

int dg16_msm(dg16_ctx* ctx, int curve, int group, const void* bases, const void* scalars,
             size_t n_bases, size_t n_scalars, unsigned flags, int channel, void* out) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    check_curve_group(curve, group);
    DG_REQUIRE(n_bases == n_scalars, DG16_ERR_LENGTH_MISMATCH,
               "bases and scalars differ in length (VariableBaseMSM::msm returns Err(min_len))");
    DG_REQUIRE(out && (n_bases == 0 || (bases && scalars)), DG16_ERR_BAD_ARG, "null operand");
    DG_REQUIRE(n_bases < ((size_t)1 << 30), DG16_ERR_BAD_ARG, "n must be < 2^30");
    bool dev = flags & DG16_F_DEVICE_PTRS;
    bool affine = flags & DG16_F_OUT_AFFINE;
    size_t pb = affine_bytes(curve, group);
    size_t ob = pb / 2 * (affine ? 2 : 3);
    Call k(ctx, channel);
    const void* dbases = stage_in(k, 0, bases, n_bases * pb, dev);
    const void* dscal = stage_in(k, 1, scalars, n_bases * 32, dev);
    void* dout = dev ? out : ws(k.c, 2, ob);
    msm_launch(k, curve, group, dbases, dscal, n_bases, msm_mode(flags), affine, dout);
    if (!dev) stage_out(k, out, dout, ob, false);
    k.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k.s()));
  });
}

int dg16_ctx_set_table_budget(dg16_ctx* ctx, uint64_t bytes) {
  if (!ctx) return DG16_ERR_BAD_ARG;
  ctx->table_budget = (size_t)bytes;
  return DG16_OK;
}

int dg16_bases_upload(dg16_ctx* ctx, int curve, int group, const void* bases, size_t n, unsigned flags,
                      dg16_bases** out) {
  if (!ctx || !out) return DG16_ERR_BAD_ARG;
  *out = nullptr;
  dg16_bases* h = new dg16_bases();
  int rc = guarded(ctx, [&] {
    check_curve_group(curve, group);
    DG_REQUIRE(bases || n == 0, DG16_ERR_BAD_ARG, "null bases");
    bool dev = flags & DG16_F_DEVICE_PTRS;
    const size_t pb = affine_bytes(curve, group);
    Call k(ctx, 0);
    const void* d = stage_in(k, 0, bases, n * pb, dev);
    h->ctx = ctx;
    h->curve = curve;
    h->group = group;
    h->n = n;
    h->table = bases_table_launch(k, curve, group, d, n, ctx->table_budget, &h->c, &h->nwin, &h->stride);
    h->bytes = (size_t)h->nwin * (n ? n : 1) * pb;
    k.finish();
    DG_HIP(hipStreamSynchronize(k.s()));     // the caller may free `bases` on return
  });
  if (rc != DG16_OK) {
    if (h->table) hipFree(h->table);
    delete h;
    return rc;
  }
  *out = h;
  return DG16_OK;
}

void dg16_bases_free(dg16_bases* h) {
  if (!h) return;
  hipSetDevice(h->ctx->device);
  hipDeviceSynchronize();
  if (h->table) hipFree(h->table);
  delete h;
}

int dg16_bases_info(const dg16_bases* h, size_t* n, unsigned* window_bits, uint64_t* table_bytes) {
  if (!h) return DG16_ERR_BAD_ARG;
  if (n) *n = h->n;
  if (window_bits) *window_bits = h->c;
  if (table_bytes) *table_bytes = h->bytes;
  return DG16_OK;
}

int dg16_msm_resident(dg16_ctx* ctx, const dg16_bases* h, const void* scalars, size_t n_scalars, unsigned flags,
                      int channel, void* out) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    DG_REQUIRE(h && h->ctx == ctx, DG16_ERR_BAD_ARG, "bases handle belongs to another context");
    DG_REQUIRE(h->n == n_scalars, DG16_ERR_LENGTH_MISMATCH,
               "bases and scalars differ in length (VariableBaseMSM::msm returns Err(min_len))");
    DG_REQUIRE(out && (scalars || n_scalars == 0), DG16_ERR_BAD_ARG, "null operand");
    bool dev = flags & DG16_F_DEVICE_PTRS, aff = flags & DG16_F_OUT_AFFINE;
    const size_t ob = affine_bytes(h->curve, h->group) / 2 * (aff ? 2 : 3);
    Call k(ctx, channel);
    const void* ds = stage_in(k, 1, scalars, n_scalars * 32, dev);
    void* dout = dev ? out : ws(k.c, 2, ob);
    msm_resident_launch(k, h->curve, h->group, h->table, h->n, h->c, h->stride, ds, flags & DG16_F_SCALARS_MONT, aff,
                        dout);
    if (!dev) stage_out(k, out, dout, ob, false);
    k.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k.s()));
  });
}

int dg16_gen_bases(dg16_ctx* ctx, int curve, int group, uint64_t seed, size_t n, void* out,
                   unsigned flags, int channel) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    check_curve_group(curve, group);
    DG_REQUIRE(out || n == 0, DG16_ERR_BAD_ARG, "null output");
    bool dev = flags & DG16_F_DEVICE_PTRS;
    size_t pb = affine_bytes(curve, group);
    Call k(ctx, channel);
    void* dout = dev ? out : ws(k.c, 0, n * pb);
    if (n) gen_bases_launch(k, curve, group, seed, n, dout);
    if (!dev) stage_out(k, out, dout, n * pb, false);
    k.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k.s()));
  });
}

int dg16_to_affine(dg16_ctx* ctx, int curve, int group, const void* jac, void* out, size_t n,
                   unsigned flags, int channel) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    check_curve_group(curve, group);
    DG_REQUIRE((jac && out) || n == 0, DG16_ERR_BAD_ARG, "null operand");
    bool dev = flags & DG16_F_DEVICE_PTRS;
    size_t pb = affine_bytes(curve, group);
    Call k(ctx, channel);
    const void* din = stage_in(k, 0, jac, n * pb / 2 * 3, dev);
    void* dout = dev ? out : ws(k.c, 1, n * pb);
    if (n) to_affine_launch(k, curve, group, din, dout, n);
    if (!dev) stage_out(k, out, dout, n * pb, false);
    k.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k.s()));
  });
}

// QAP evaluation vectors from the R1CS matrices and the full assignment.
int dg16_qap(dg16_ctx* ctx, int curve, size_t num_constraints, size_t num_inputs, size_t num_vars, unsigned log_m,
             const uint32_t* a_row_ptr, const uint32_t* a_col, const void* a_coeff, const uint32_t* b_row_ptr,
             const uint32_t* b_col, const void* b_coeff, const void* full_assignment, void* a_out, void* b_out,
             void* c_out, unsigned flags, int channel) {
  return dg16_qap_rows(ctx, curve, num_constraints, num_inputs, num_vars, log_m, a_row_ptr, a_col, a_coeff, b_row_ptr,
                       b_col, b_coeff, full_assignment, 0, 1, a_out, b_out, c_out, flags, channel);
}

int dg16_qap_rows(dg16_ctx* ctx, int curve, size_t num_constraints, size_t num_inputs, size_t num_vars, unsigned log_m,
                  const uint32_t* a_row_ptr, const uint32_t* a_col, const void* a_coeff, const uint32_t* b_row_ptr,
                  const uint32_t* b_col, const void* b_coeff, const void* full_assignment, size_t row_start,
                  size_t row_stride, void* a_out, void* b_out, void* c_out, unsigned flags, int channel) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    DG_REQUIRE(curve >= 0 && curve <= 2, DG16_ERR_BAD_CURVE, "unknown curve id");
    DG_REQUIRE(a_row_ptr && b_row_ptr && full_assignment && a_out && b_out && c_out, DG16_ERR_BAD_ARG, "null operand");
    const size_t m = (size_t)1 << log_m;
    DG_REQUIRE(row_stride >= 1 && !(row_stride & (row_stride - 1)) && row_stride <= m && row_start < row_stride,
               DG16_ERR_BAD_ARG, "row_stride must be a power of two <= the domain, row_start < row_stride");
    const size_t rows = m / row_stride;     // output elements per vector
    // D::new(num_constraints + num_inputs) (qap.rs:53): the domain must hold both
    DG_REQUIRE(num_constraints + num_inputs <= m && num_inputs <= num_vars, DG16_ERR_BAD_ARG,
               "domain smaller than num_constraints + num_inputs");
    bool dev = flags & DG16_F_DEVICE_PTRS;
    Call k(ctx, channel, /*tail_ok=*/dev);      // on device pointers this call touches no workspace (ctx.h: tail_pending)
    // Matrix indices are checked before they index the assignment: a malformed key file must end in
    // DG16_ERR_BAD_ARG, not in an out-of-bounds device read (the reference's Rust indexing panics).  Host
    // pointers: checked here.  Device pointers: no read-back (the call stays stream-ordered); the kernel skips
    // every out-of-range entry and raises the context's sticky device flag, which the next dg16_sync reports.
    size_t a_nnz = 0, b_nnz = 0;
    if (!dev) {
      auto check = [&](const uint32_t* ptr, const uint32_t* col, size_t& nnz) {
        DG_REQUIRE(ptr[0] == 0, DG16_ERR_BAD_ARG, "row_ptr[0] != 0");
        for (size_t i = 0; i < num_constraints; i++)
          DG_REQUIRE(ptr[i] <= ptr[i + 1], DG16_ERR_BAD_ARG, "row_ptr is not monotonic");
        nnz = ptr[num_constraints];
        DG_REQUIRE(nnz == 0 || col, DG16_ERR_BAD_ARG, "null column indices");
        for (size_t j = 0; j < nnz; j++)
          DG_REQUIRE(col[j] < num_vars, DG16_ERR_BAD_ARG, "coefficient out of range (column >= num_vars)");
      };
      check(a_row_ptr, a_col, a_nnz);
      check(b_row_ptr, b_col, b_nnz);
      DG_REQUIRE((a_nnz == 0 || a_coeff) && (b_nnz == 0 || b_coeff), DG16_ERR_BAD_ARG, "null coefficients");
    }
    const unsigned* ap = (const unsigned*)stage_in(k, 0, a_row_ptr, (num_constraints + 1) * 4, dev);
    const unsigned* ac = (const unsigned*)stage_in(k, 1, a_col, a_nnz * 4, dev);
    const void* av = stage_in(k, 2, a_coeff, a_nnz * 32, dev);
    const unsigned* bp = (const unsigned*)stage_in(k, 3, b_row_ptr, (num_constraints + 1) * 4, dev);
    const unsigned* bc = (const unsigned*)stage_in(k, 18, b_col, b_nnz * 4, dev);
    const void* bv = stage_in(k, 19, b_coeff, b_nnz * 32, dev);
    const void* w = stage_in(k, 20, full_assignment, num_vars * 32, dev);
    uint8_t* out = dev ? nullptr : (uint8_t*)ws(k.c, 21, 3 * rows * 32);
    void* da = dev ? a_out : out;
    void* db = dev ? b_out : out + rows * 32;
    void* dc = dev ? c_out : out + 2 * rows * 32;
    qap_launch(k, curve, ap, ac, av, bp, bc, bv, w, flags & DG16_F_SCALARS_MONT, num_constraints, num_inputs, num_vars,
               m, row_start, row_stride, da, db, dc);
    if (!dev) {
      stage_out(k, a_out, da, rows * 32, false);
      stage_out(k, b_out, db, rows * 32, false);
      stage_out(k, c_out, dc, rows * 32, false);
    }
    k.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k.s()));
  });
}


int dg16_ntt_dist_stage(dg16_ctx* ctx, int curve, unsigned log_n, unsigned rank, unsigned n_ranks, int inverse,
                        int stage, const void* in, void* out, unsigned flags, int channel) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    DG_REQUIRE(curve >= 0 && curve <= 2, DG16_ERR_BAD_CURVE, "unknown curve id");
    DG_REQUIRE(flags & DG16_F_DEVICE_PTRS, DG16_ERR_BAD_ARG, "the sharded NTT works on device buffers");
    DG_REQUIRE(in && out && (stage == 0 || stage == 1), DG16_ERR_BAD_ARG, "null operand or unknown stage");
    Call k(ctx, channel);
    k.begin_dominant();
    ntt_dist_stage(k, curve, log_n, rank, n_ranks, inverse, stage, in, out);
    k.end_dominant();
    k.finish();
  });
}

int dg16_ntt_dist(dg16_ctx* ctx, int curve, const dg16_comm* comm, const void* in, void* out, unsigned log_n,
                  int inverse, unsigned flags, int channel) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    DG_REQUIRE(curve >= 0 && curve <= 2, DG16_ERR_BAD_CURVE, "unknown curve id");
    DG_REQUIRE(flags & DG16_F_DEVICE_PTRS, DG16_ERR_BAD_ARG, "the sharded NTT works on device buffers");
    DG_REQUIRE(in && out, DG16_ERR_BAD_ARG, "null operand");
    Call k(ctx, channel);
    k.begin_dominant();
    if (!comm || comm->n_ranks(comm->self) == 1) {
      if (in != out) DG_HIP(hipMemcpyAsync(out, in, (size_t)32 << log_n, hipMemcpyDeviceToDevice, k.s()));
      ntt_launch(k, curve, out, log_n, inverse, nullptr);
    } else {
      ntt_dist_launch(k, curve, comm, in, out, log_n, inverse);
    }
    k.end_dominant();
    k.finish();
  });
}

int dg16_h_poly_dist_stage(dg16_ctx* ctx, int curve, unsigned log_m, unsigned rank, unsigned n_ranks, int stage,
                           const void* const* in, void* out, unsigned flags, int channel) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    DG_REQUIRE(curve >= 0 && curve <= 2, DG16_ERR_BAD_CURVE, "unknown curve id");
    DG_REQUIRE(flags & DG16_F_DEVICE_PTRS, DG16_ERR_BAD_ARG, "the sharded h-polynomial works on device buffers");
    DG_REQUIRE(in && in[0] && out && stage >= 0 && stage <= 2 && (stage != 0 || (in[1] && in[2])), DG16_ERR_BAD_ARG,
               "null operand or unknown stage");
    Call k(ctx, channel);
    k.begin_dominant();
    h_poly_dist_stage(k, curve, log_m, rank, n_ranks, stage, in, out);
    k.end_dominant();
    k.finish();
  });
}

int dg16_h_poly_dist(dg16_ctx* ctx, int curve, const dg16_comm* comm, const void* a_rows, const void* b_rows,
                     const void* c_rows, unsigned log_m, void* out, unsigned flags, int channel) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    DG_REQUIRE(curve >= 0 && curve <= 2, DG16_ERR_BAD_CURVE, "unknown curve id");
    DG_REQUIRE(flags & DG16_F_DEVICE_PTRS, DG16_ERR_BAD_ARG, "the sharded h-polynomial works on device buffers");
    DG_REQUIRE(a_rows && b_rows && c_rows && out, DG16_ERR_BAD_ARG, "null operand");
    Call k(ctx, channel);
    k.begin_dominant();
    if (!comm || comm->n_ranks(comm->self) == 1)
      h_poly_launch(k, curve, a_rows, b_rows, c_rows, log_m, out);
    else
      h_poly_dist_launch(k, curve, comm, a_rows, b_rows, c_rows, log_m, out);
    k.end_dominant();
    k.finish();
  });
}

}  // extern "C"
