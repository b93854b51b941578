// arkworks compressed key files for BN254 (and the point codec for BLS12-377 and BLS12-381 too) (`ProvingKey::<Bn254>` / `VerifyingKey::<Bn254>` written with
// `serialize_with_mode(.., Compress::Yes)` and read back per request with `deserialize_with_mode(.., Compress::Yes,
// Validate::No)`: mpc-api/src/main.rs:154-171, :459-512).
//
// Two layers:
//   dg16_arkkey_layout      host code: walks the container -- field order of the derive macro, u64 little-endian
//                           lengths in front of every Vec -- and returns counts and byte offsets
//   dg16_points_compress /  the batched point codec on the GPU, one lane per point: the SAME encode / decode routines
//   dg16_points_decompress  as the proof.bin codec (codec_impl.h, pinned by the reference's own proof.bin), so a
//                           2^20-point query decompresses (one 254-bit exponentiation per G1 point for the square
//                           root, a handful per G2 point) in milliseconds instead of a minute of host time, straight
//                           into the buffers dg16_pk_create / dg16_bases_upload take
// Layout of the files (ark-groth16 0.4 `data_structures.rs`, struct field order):
//   VerifyingKey: alpha_g1 (32) | beta_g2 (64) | gamma_g2 (64) | delta_g2 (64) | len u64 | gamma_abc_g1[len] (32 each)
//   ProvingKey:   VerifyingKey | beta_g1 (32) | delta_g1 (32) | len | a_query | len | b_g1_query | len | b_g2_query (64
//                 each) | len | h_query | len | l_query
#include "codec_impl.h"
#include "ctx.h"

namespace dg16 {

template <int CURVE, class F>
__global__ void __launch_bounds__(64) points_encode_kernel(const Affine<F>* __restrict__ in, size_t n,
                                                            uint8_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr size_t CB = sizeof(F);      // compressed size = one coordinate
  uint8_t buf[CB];
  CodecT<CURVE>::encode(in[i], buf);
  for (size_t k = 0; k < CB; k++) out[i * CB + k] = buf[k];
}

// err[0]: smallest failing error code's first index + 1 (0 = none), err[1]: its code
__device__ void report(unsigned* err, size_t i, int code) {
  const unsigned long long tag = ((unsigned long long)(i + 1) << 8) | (unsigned)code;
  atomicMin((unsigned long long*)err, tag);
}
template <int CURVE, class F>
__global__ void __launch_bounds__(64) points_decode_kernel(const uint8_t* __restrict__ in, size_t n, int validate,
                                                            Affine<F>* __restrict__ out, unsigned* err) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr size_t CB = sizeof(F);
  uint8_t buf[CB];
  for (size_t k = 0; k < CB; k++) buf[k] = in[i * CB + k];
  Affine<F> p;
  const int rc = CodecT<CURVE>::decode(buf, p, validate != 0);
  if (rc) { report(err, i, rc); p = Affine<F>::inf(); }
  out[i] = p;
}

template <int CURVE>
void launch_encode(hipStream_t s, int group, const void* din, size_t n, uint8_t* dout) {
  using C = CodecT<CURVE>;
  const unsigned blocks = (unsigned)((n + 63) / 64);
  if (group == 1)
    hipLaunchKernelGGL((points_encode_kernel<CURVE, typename C::Fq>), dim3(blocks), dim3(64), 0, s,
                       (const Affine<typename C::Fq>*)din, n, dout);
  else
    hipLaunchKernelGGL((points_encode_kernel<CURVE, typename C::Fq2>), dim3(blocks), dim3(64), 0, s,
                       (const Affine<typename C::Fq2>*)din, n, dout);
}
template <int CURVE>
void launch_decode(hipStream_t s, int group, const uint8_t* din, size_t n, int validate, void* dout, unsigned* err) {
  using C = CodecT<CURVE>;
  const unsigned blocks = (unsigned)((n + 63) / 64);
  if (group == 1)
    hipLaunchKernelGGL((points_decode_kernel<CURVE, typename C::Fq>), dim3(blocks), dim3(64), 0, s, din, n, validate,
                       (Affine<typename C::Fq>*)dout, err);
  else
    hipLaunchKernelGGL((points_decode_kernel<CURVE, typename C::Fq2>), dim3(blocks), dim3(64), 0, s, din, n, validate,
                       (Affine<typename C::Fq2>*)dout, err);
}

// ---- Vec<F> on the wire: ark-serialize compressed form = u64 length || canonical little-endian elements ----------
template <class Fr>
__global__ void __launch_bounds__(256) wire_fr_encode_kernel(const Fr* __restrict__ in, size_t n, uint32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fr c = in[i].from_mont();
#pragma unroll
  for (int k = 0; k < Fr::NL; k++) out[i * Fr::NL + k] = c.l[k];
}
template <class Fr>
__global__ void __launch_bounds__(256) wire_fr_decode_kernel(const uint32_t* __restrict__ in, size_t n, Fr* __restrict__ out,
                                                              unsigned* err) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr c;
#pragma unroll
  for (int k = 0; k < Fr::NL; k++) c.l[k] = in[i * Fr::NL + k];
  bool lt = false;                       // canonical: c < r
  for (int k = Fr::NL - 1; k >= 0; k--)
    if (c.l[k] != Fr::Params::P[k]) { lt = c.l[k] < Fr::Params::P[k]; break; }
  if (!lt) { report(err, i, 2); c = Fr::zero(); }
  out[i] = c.to_mont();
}

}  // namespace dg16

using namespace dg16;

namespace {
thread_local std::string g_codec_err;
const char* kCodecErr[5] = {"", "invalid flags", "coordinate not reduced", "x is not on the curve",
                            "point is not in the prime-order subgroup"};
uint64_t rd64(const uint8_t* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}
}  // namespace

extern "C" {

const char* dg16_codec_error(void) { return g_codec_err.c_str(); }

int dg16_points_compress(dg16_ctx* ctx, int curve, int group, const void* affine, size_t n, void* out, unsigned flags,
                         int channel) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    DG_REQUIRE(curve >= 0 && curve <= 2, DG16_ERR_BAD_CURVE, "unknown curve id");
    DG_REQUIRE(group == 1 || group == 2, DG16_ERR_BAD_ARG, "group must be 1 (G1) or 2 (G2)");
    DG_REQUIRE((affine && out) || n == 0, DG16_ERR_BAD_ARG, "null operand");
    const bool dev = flags & DG16_F_DEVICE_PTRS;
    const size_t fb = curve == DG16_BN254 ? 32 : 48, pb = 2 * fb * group, cb = fb * group;
    Call k(ctx, channel);
    const void* din = stage_in(k, 0, affine, n * pb, dev);
    uint8_t* dout = dev ? (uint8_t*)out : (uint8_t*)ws(k.c, 1, n * cb);
    if (n) {
      if (curve == DG16_BN254) launch_encode<0>(k.s(), group, din, n, dout);
      else if (curve == DG16_BLS12_381) launch_encode<1>(k.s(), group, din, n, dout);
      else launch_encode<2>(k.s(), group, din, n, dout);
      DG_HIP(hipGetLastError());
    }
    if (!dev) stage_out(k, out, dout, n * cb, false);
    k.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k.s()));
  });
}

// Synchronous also with device pointers: the outcome (is every x on the curve?) is part of the return value, like
// the Err of deserialize_with_mode.
int dg16_points_decompress(dg16_ctx* ctx, int curve, int group, const void* in, size_t n, int validate,
                           void* affine_out, unsigned flags, int channel) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    DG_REQUIRE(curve >= 0 && curve <= 2, DG16_ERR_BAD_CURVE, "unknown curve id");
    DG_REQUIRE(group == 1 || group == 2, DG16_ERR_BAD_ARG, "group must be 1 (G1) or 2 (G2)");
    DG_REQUIRE((in && affine_out) || n == 0, DG16_ERR_BAD_ARG, "null operand");
    const bool dev = flags & DG16_F_DEVICE_PTRS;
    const size_t fb = curve == DG16_BN254 ? 32 : 48, pb = 2 * fb * group, cb = fb * group;
    Call k(ctx, channel);
    const uint8_t* din = (const uint8_t*)stage_in(k, 0, in, n * cb, dev);
    void* dout = dev ? affine_out : ws(k.c, 1, n * pb);
    unsigned long long* err = (unsigned long long*)ws(k.c, 2, 16);
    DG_HIP(hipMemsetAsync(err, 0xFF, 8, k.s()));
    if (n) {
      if (curve == DG16_BN254) launch_decode<0>(k.s(), group, din, n, validate, dout, (unsigned*)err);
      else if (curve == DG16_BLS12_381) launch_decode<1>(k.s(), group, din, n, validate, dout, (unsigned*)err);
      else launch_decode<2>(k.s(), group, din, n, validate, dout, (unsigned*)err);
      DG_HIP(hipGetLastError());
    }
    if (!dev) stage_out(k, affine_out, dout, n * pb, false);
    unsigned long long tag = ~0ull;
    DG_HIP(hipMemcpyAsync(&tag, err, 8, hipMemcpyDeviceToHost, k.s()));
    k.finish();
    DG_HIP(hipStreamSynchronize(k.s()));
    if (tag != ~0ull) {
      const unsigned code = (unsigned)(tag & 0xFF);
      g_codec_err = std::string(kCodecErr[code < 5 ? code : 0]) + " (point " + std::to_string((tag >> 8) - 1) + ")";
      throw StatusError{DG16_ERR_BAD_ARG, g_codec_err};
    }
  });
}

size_t dg16_wire_fr_bytes(size_t n) { return 8 + 32 * n; }

int dg16_wire_fr_encode(dg16_ctx* ctx, int curve, const void* mont, size_t n, void* out, unsigned flags, int channel) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    DG_REQUIRE(curve >= 0 && curve <= 2, DG16_ERR_BAD_CURVE, "unknown curve id");
    DG_REQUIRE(out && (mont || n == 0), DG16_ERR_BAD_ARG, "null operand");
    const bool dev = flags & DG16_F_DEVICE_PTRS;
    Call k(ctx, channel);
    const void* din = stage_in(k, 0, mont, n * 32, dev);
    uint8_t* dout = dev ? (uint8_t*)out : (uint8_t*)ws(k.c, 1, 8 + n * 32);
    const uint64_t len = n;
    DG_HIP(hipMemcpyAsync(dout, &len, 8, hipMemcpyHostToDevice, k.s()));
    DG_HIP(hipStreamSynchronize(k.s()));                     // `len` lives on this stack frame
    if (n) {
      const unsigned blocks = (unsigned)((n + 255) / 256);
      uint32_t* body = (uint32_t*)(dout + 8);
      switch (curve) {
        case 0: hipLaunchKernelGGL(wire_fr_encode_kernel<bn254_fr>, dim3(blocks), dim3(256), 0, k.s(), (const bn254_fr*)din, n, body); break;
        case 1: hipLaunchKernelGGL(wire_fr_encode_kernel<bls12_381_fr>, dim3(blocks), dim3(256), 0, k.s(), (const bls12_381_fr*)din, n, body); break;
        default: hipLaunchKernelGGL(wire_fr_encode_kernel<bls12_377_fr>, dim3(blocks), dim3(256), 0, k.s(), (const bls12_377_fr*)din, n, body); break;
      }
      DG_HIP(hipGetLastError());
    }
    if (!dev) stage_out(k, out, dout, 8 + n * 32, false);
    k.finish();
    if (!dev) DG_HIP(hipStreamSynchronize(k.s()));
  });
}

// Synchronous (the length prefix and the "< r" check decide the return value, like deserialize_compressed's Err).
int dg16_wire_fr_decode(dg16_ctx* ctx, int curve, const void* in, size_t bytes, void* out_mont, size_t* n_out,
                        unsigned flags, int channel) {
  int rc = guard_channel(ctx, channel);
  if (rc) return rc;
  return guarded(ctx, [&] {
    DG_REQUIRE(curve >= 0 && curve <= 2, DG16_ERR_BAD_CURVE, "unknown curve id");
    DG_REQUIRE(in && n_out && bytes >= 8, DG16_ERR_BAD_ARG, "null operand or missing length prefix");
    const bool dev = flags & DG16_F_DEVICE_PTRS;
    Call k(ctx, channel);
    const uint8_t* din = (const uint8_t*)stage_in(k, 0, in, bytes, dev);
    uint64_t len = 0;
    DG_HIP(hipMemcpyAsync(&len, din, 8, hipMemcpyDeviceToHost, k.s()));
    DG_HIP(hipStreamSynchronize(k.s()));
    if (len != (bytes - 8) / 32 || (bytes - 8) % 32) {
      g_codec_err = "Vec<F>: length prefix does not match the payload";
      throw StatusError{DG16_ERR_BAD_ARG, g_codec_err};
    }
    const size_t n = (size_t)len;
    *n_out = n;
    DG_REQUIRE(out_mont || n == 0, DG16_ERR_BAD_ARG, "null output");
    void* dout = dev ? out_mont : ws(k.c, 1, n * 32);
    unsigned long long* err = (unsigned long long*)ws(k.c, 2, 16);
    DG_HIP(hipMemsetAsync(err, 0xFF, 8, k.s()));
    if (n) {
      const unsigned blocks = (unsigned)((n + 255) / 256);
      const uint32_t* body = (const uint32_t*)(din + 8);
      switch (curve) {
        case 0: hipLaunchKernelGGL(wire_fr_decode_kernel<bn254_fr>, dim3(blocks), dim3(256), 0, k.s(), body, n, (bn254_fr*)dout, (unsigned*)err); break;
        case 1: hipLaunchKernelGGL(wire_fr_decode_kernel<bls12_381_fr>, dim3(blocks), dim3(256), 0, k.s(), body, n, (bls12_381_fr*)dout, (unsigned*)err); break;
        default: hipLaunchKernelGGL(wire_fr_decode_kernel<bls12_377_fr>, dim3(blocks), dim3(256), 0, k.s(), body, n, (bls12_377_fr*)dout, (unsigned*)err); break;
      }
      DG_HIP(hipGetLastError());
    }
    if (!dev) stage_out(k, out_mont, dout, n * 32, false);
    unsigned long long tag = ~0ull;
    DG_HIP(hipMemcpyAsync(&tag, err, 8, hipMemcpyDeviceToHost, k.s()));
    k.finish();
    DG_HIP(hipStreamSynchronize(k.s()));
    if (tag != ~0ull) {
      g_codec_err = "Vec<F>: element " + std::to_string((tag >> 8) - 1) + " is not reduced";
      throw StatusError{DG16_ERR_BAD_ARG, g_codec_err};
    }
  });
}

int dg16_arkkey_layout(const void* data, size_t bytes, int verifying_key_only, dg16_arkkey_layout_t* out) {
  if (!data || !out) return DG16_ERR_BAD_ARG;
  memset(out, 0, sizeof(*out));
  const uint8_t* p = (const uint8_t*)data;
  size_t at = 0;
  auto need = [&](size_t k) {
    if (at + k > bytes || at + k < at) { g_codec_err = "key file truncated"; return false; }
    return true;
  };
  auto fixed = [&](uint64_t& off, size_t k) {
    if (!need(k)) return false;
    off = at;
    at += k;
    return true;
  };
  auto vec = [&](uint64_t& off, uint64_t& count, size_t each) {
    if (!need(8)) return false;
    count = rd64(p + at);
    at += 8;
    if (count > (bytes - at) / each) { g_codec_err = "key file truncated (vector length exceeds the file)"; return false; }
    off = at;
    at += (size_t)count * each;
    return true;
  };
  bool ok = fixed(out->off_alpha_g1, 32) && fixed(out->off_beta_g2, 64) && fixed(out->off_gamma_g2, 64) &&
            fixed(out->off_delta_g2, 64) && vec(out->off_ic, out->n_ic, 32);
  if (ok && !verifying_key_only)
    ok = fixed(out->off_beta_g1, 32) && fixed(out->off_delta_g1, 32) && vec(out->off_a, out->n_a, 32) &&
         vec(out->off_b1, out->n_b1, 32) && vec(out->off_b2, out->n_b2, 64) && vec(out->off_h, out->n_h, 32) &&
         vec(out->off_l, out->n_l, 32);
  if (!ok) return DG16_ERR_BAD_ARG;
  if (at != bytes) { g_codec_err = "trailing bytes after the key"; return DG16_ERR_BAD_ARG; }
  out->bytes = at;
  return DG16_OK;
}

}  // extern "C"
