// arkworks compressed Proof<Bn254> <-> the prover's output -- host side of libdg16 (no GPU work here).  The
// reference writes / reads this 128-byte encoding as `proof.bin` (`proof.serialize_compressed`,
// `Proof::deserialize_compressed`: mpc-api/src/main.rs:154-171, zk-cli/test-circuits/sha256/proof.bin):
// A (G1, 32 B) || B (G2, 64 B) || C (G1, 32 B), little-endian x with the ark-ec short-Weierstrass flags in the
// two top bits of the last byte (bit 7: y is the "negative" root, y > -y, for Fq2 compared on (c1, c0);
// bit 6: point at infinity).  Pinned by the reference's own proof.bin and the coordinates its CLI prints
// (tests/test_serialize.py).  Uses the portable host path of fp.h.
#include <stdint.h>
#include <string.h>

#include "../../include/dg16.h"
#include "codec_impl.h"

namespace {

using namespace dg16;
using namespace dg16::codec;

thread_local const char* g_err = "";
const char* kDecodeErr[5] = {"", "invalid flags", "coordinate not reduced", "x is not on the curve",
                             "point is not in the prime-order subgroup"};

}  // namespace

extern "C" {

const char* dg16_serialize_error(void) { return g_err; }

int dg16_proof_compress(int curve, const void* proof_jacobian, void* out) {
  if (curve != DG16_BN254) { g_err = "compressed proofs: BN254 only"; return DG16_ERR_UNSUPPORTED; }
  if (!proof_jacobian || !out) { g_err = "null argument"; return DG16_ERR_BAD_ARG; }
  const uint8_t* in = (const uint8_t*)proof_jacobian;
  Jacobian<Fq> a, c;
  Jacobian<Fq2> b;
  memcpy(&a, in, sizeof(a));
  memcpy(&b, in + sizeof(a), sizeof(b));
  memcpy(&c, in + sizeof(a) + sizeof(b), sizeof(c));
  uint8_t* o = (uint8_t*)out;
  encode(XYZZ<Fq>::from_jacobian(a).to_affine(), o);
  encode(XYZZ<Fq2>::from_jacobian(b).to_affine(), o + 32);
  encode(XYZZ<Fq>::from_jacobian(c).to_affine(), o + 96);
  return DG16_OK;
}

int dg16_proof_decompress(int curve, const void* in128, int validate, void* proof_affine) {
  if (curve != DG16_BN254) { g_err = "compressed proofs: BN254 only"; return DG16_ERR_UNSUPPORTED; }
  if (!in128 || !proof_affine) { g_err = "null argument"; return DG16_ERR_BAD_ARG; }
  const uint8_t* in = (const uint8_t*)in128;
  Affine<Fq> a, c;
  Affine<Fq2> b;
  int rc = decode(in, a);
  if (!rc) rc = decode(in + 32, b, validate != 0);
  if (!rc) rc = decode(in + 96, c);
  if (rc) { g_err = kDecodeErr[rc]; return DG16_ERR_BAD_ARG; }
  uint8_t* o = (uint8_t*)proof_affine;
  memcpy(o, &a, sizeof(a));
  memcpy(o + sizeof(a), &b, sizeof(b));
  memcpy(o + sizeof(a) + sizeof(b), &c, sizeof(c));
  return DG16_OK;
}

}  // extern "C"
