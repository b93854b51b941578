// Test-only: instantiates the product's field / curve headers (distributed-groth16_amd/csrc/fp.h,
// fp2.h, ec.h) with the HOST compiler so the formulas can be checked against the oracle without a
// GPU (-m "not gpu").  The portable C++ multiply path runs here; the gfx950 asm path is checked on
// the GPU by tests/test_gpu_field.py.  Never part of the product.
#include <stdint.h>
#include <string.h>
#include "../../distributed-groth16_amd/csrc/consts_gen.h"
#include "../../distributed-groth16_amd/csrc/ec.h"
#include "../../distributed-groth16_amd/csrc/ec29.h"
#include "../../distributed-groth16_amd/csrc/glv.h"

using namespace dg16;

template <class F> static void field_op(int op, const F* a, const F* b, F* o, size_t n) {
  for (size_t i = 0; i < n; i++) {
    switch (op) {
      case 0: o[i] = a[i] + b[i]; break;
      case 1: o[i] = a[i] - b[i]; break;
      case 2: o[i] = a[i] * b[i]; break;
      case 3: o[i] = a[i].sqr(); break;
      case 4: o[i] = a[i].inv(); break;
      case 5: o[i] = a[i].to_mont(); break;
      case 6: o[i] = a[i].from_mont(); break;
      case 7: o[i] = a[i].neg(); break;
    }
  }
}

// op: 0 = xyzz(a).add(xyzz(b)) via general add, 1 = madd(a, b), 2 = madd(a, -b), 3 = dbl(a),
//     4 = scalar_mul(a, k[8 words] in b's first 32 bytes); result -> affine
template <class F> static void point_op(int op, const Affine<F>* a, const Affine<F>* b, Affine<F>* o, size_t n) {
  for (size_t i = 0; i < n; i++) {
    XYZZ<F> A = XYZZ<F>::from_affine(a[i]);
    XYZZ<F> r;
    switch (op) {
      case 0: {
        // go through a non-trivial ZZ by doubling-and-subtracting first so add-2008-s sees z != 1
        XYZZ<F> B = XYZZ<F>::from_affine(b[i]);
        XYZZ<F> A3 = A.dbl().add(A);          // 3a
        XYZZ<F> B3 = B.dbl().add(B);          // 3b
        r = A3.add(B3).add(A.dbl().neg()).add(B.dbl().neg());  // 3a + 3b - 2a - 2b
        break;
      }
      case 1: r = A.dbl().madd(b[i], false).add(A.neg()); break;     // 2a + b - a
      case 2: r = A.madd(b[i], true); break;
      case 3: r = A.dbl(); break;
      default: r = scalar_mul<F, 8>(A, (const uint32_t*)&b[i]); break;
    }
    Jacobian<F> j = r.to_jacobian();
    o[i] = XYZZ<F>::from_jacobian(j).to_affine();
  }
}

using bn_fq = Fp<bn254_fq_params>;      using bn_fr = Fp<bn254_fr_params>;
using b381_fq = Fp<bls12_381_fq_params>; using b381_fr = Fp<bls12_381_fr_params>;
using b377_fq = Fp<bls12_377_fq_params>; using b377_fr = Fp<bls12_377_fr_params>;

extern "C" int ha_field_op(int fid, int op, const void* a, const void* b, void* o, size_t n) {
  switch (fid) {
    case 0: field_op<bn_fq>(op, (const bn_fq*)a, (const bn_fq*)b, (bn_fq*)o, n); break;
    case 1: field_op<b381_fq>(op, (const b381_fq*)a, (const b381_fq*)b, (b381_fq*)o, n); break;
    case 2: field_op<b377_fq>(op, (const b377_fq*)a, (const b377_fq*)b, (b377_fq*)o, n); break;
    case 16: field_op<bn_fr>(op, (const bn_fr*)a, (const bn_fr*)b, (bn_fr*)o, n); break;
    case 17: field_op<b381_fr>(op, (const b381_fr*)a, (const b381_fr*)b, (b381_fr*)o, n); break;
    case 18: field_op<b377_fr>(op, (const b377_fr*)a, (const b377_fr*)b, (b377_fr*)o, n); break;
    default: return 1;
  }
  return 0;
}
extern "C" int ha_point_op(int curve, int group, int op, const void* a, const void* b, void* o, size_t n) {
  switch (curve * 2 + group - 1) {
    case 0: point_op<bn_fq>(op, (const Affine<bn_fq>*)a, (const Affine<bn_fq>*)b, (Affine<bn_fq>*)o, n); break;
    case 1: point_op<Fp2<bn_fq>>(op, (const Affine<Fp2<bn_fq>>*)a, (const Affine<Fp2<bn_fq>>*)b, (Affine<Fp2<bn_fq>>*)o, n); break;
    case 2: point_op<b381_fq>(op, (const Affine<b381_fq>*)a, (const Affine<b381_fq>*)b, (Affine<b381_fq>*)o, n); break;
    case 3: point_op<Fp2<b381_fq>>(op, (const Affine<Fp2<b381_fq>>*)a, (const Affine<Fp2<b381_fq>>*)b, (Affine<Fp2<b381_fq>>*)o, n); break;
    case 4: point_op<b377_fq>(op, (const Affine<b377_fq>*)a, (const Affine<b377_fq>*)b, (Affine<b377_fq>*)o, n); break;
    case 5: point_op<Fp2<b377_fq>>(op, (const Affine<Fp2<b377_fq>>*)a, (const Affine<Fp2<b377_fq>>*)b, (Affine<Fp2<b377_fq>>*)o, n); break;
    default: return 1;
  }
  return 0;
}


// ---- reduced-radix types (fp29.h / ec29.h): same operations, through the 29/28-bit-limb representation ------------
// field: op 0 add, 1 sub, 2 mul, 3 sqr, 7 neg, 9 = mul_sub / mul_add4 on loose operands, 8 = mul through a long lazy chain ((a + b)(a - b) + a b - b^2 == a^2 - 2b^2 + ab)
template <class P> static void field_op29(int op, const Fp<P>* a, const Fp<P>* b, Fp<P>* o, size_t n) {
  for (size_t i = 0; i < n; i++) {
    const auto x = fe_from_fp(a[i]);
    const auto y = fe_from_fp(b[i]);
    switch (op) {
      case 0: o[i] = fe_to_fp(x + y); break;
      case 1: o[i] = fe_to_fp(x - y); break;
      case 2: o[i] = fe_to_fp(x * y); break;
      case 3: o[i] = fe_to_fp(sqr(x)); break;
      case 7: o[i] = fe_to_fp(neg(x)); break;
      case 9: {   // the fused forms on operands with loose limbs: (x + y)(x - y) - (y + y + x) x, once as mul_sub (one
                  // reduction for the 9-limb fields), once as the four-product sum with explicit negations; must agree
        const auto u = x + y;                 // limbs < 2 2^W
        const auto v = x - y;                 // limbs < 3 2^W
        const auto w = y + y + x;             // limbs < 3 2^W
        const auto r1 = mul_sub(u, v, w, x);
        const auto r2 = mul_add4(u, x, u, neg(y), neg(w), x, fe_one<P>() - fe_one<P>(), y);   // u x - u y - w x + 0 y
        o[i] = fe_to_fp(r1);
        if (!is_zero(norm(r1 - r2))) o[i].l[0] ^= 0xbeef;
        break;
      }
      default: {
        const auto t = norm(norm((x + y) * (x - y) + x * y) - sqr(y));
        const auto t4 = norm(dbl(dbl(t)));
        const auto u = reduce(t4 + t);                   // 5 t through reduce()
        o[i] = fe_to_fp(u - t4);                         // == t
        // is_zero must see through every representation of zero
        if (!is_zero(t - t) || !is_zero(norm(dbl(x) - x) - x) || is_zero(fe_one<P>() + (t - t))) o[i].l[0] ^= 0xdead;
        break;
      }
    }
  }
}
// points: the operations of point_op above on XYZZ29
template <class F> static void point_op29(int op, const Affine<F>* a, const Affine<F>* b, Affine<F>* o, size_t n) {
  using FO = FieldOf<F>;
  for (size_t i = 0; i < n; i++) {
    uint32_t wa[2 * FO::WORDS], wb[2 * FO::WORDS];
    affine_to_internal(a[i], wa);
    affine_to_internal(b[i], wb);
    const Affine29<F> pa = Affine29<F>::load(wa), pb = Affine29<F>::load(wb);
    XYZZ29<F> A = XYZZ29<F>::inf().madd(pa, false);
    XYZZ29<F> r;
    switch (op) {
      case 0: {
        XYZZ29<F> B = XYZZ29<F>::inf().madd(pb, false);
        XYZZ29<F> A3 = A.dbl_pt().add(A), B3 = B.dbl_pt().add(B);
        r = A3.add(B3).add(A.dbl_pt().neg_pt()).add(B.dbl_pt().neg_pt());
        // the memory-operand forms (reduction kernels), incl. aliased destinations, must agree
        XYZZ29<F> m3, t2 = A;
        XYZZ29<F>::dbl_mem(&t2, &t2);                 // 2a (in place)
        XYZZ29<F>::add_mem(&m3, &t2, &A);             // 3a
        XYZZ29<F> n3 = B;
        XYZZ29<F>::dbl_mem(&n3, &B);
        XYZZ29<F>::add_mem(&n3, &n3, &B);             // 3b (dst aliases a)
        XYZZ29<F>::add_mem(&n3, &m3, &n3);            // 3a + 3b (dst aliases b)
        XYZZ29<F> na = A.dbl_pt().neg_pt(), nb = B.dbl_pt().neg_pt();
        XYZZ29<F>::add_mem(&n3, &n3, &na);
        XYZZ29<F>::add_mem(&n3, &n3, &nb);
        // the accessor forms (in-workgroup bucket tree: add_acc; throughput finalize: add_into), incl. identity operands,
        // equal operands (doubling) and inverse operands (identity result): 3a + 3b - 2a - 2b again, then a + a, a - a
        {
          struct Acc {
            XYZZ29<F>* p;
            typename FO::Store get(int c) const { return c == 0 ? p->x : c == 1 ? p->y : c == 2 ? p->zz : p->zzz; }
            void put(int c, const typename FO::Store& v) const { (c == 0 ? p->x : c == 1 ? p->y : c == 2 ? p->zz : p->zzz) = v; }
          };
          for (int form = 0; form < 2; form++) {
            auto plus = [&](XYZZ29<F>& d, XYZZ29<F> o) {
              if (form == 0) XYZZ29<F>::add_acc(Acc{&d}, Acc{&d}, Acc{&o});
              else XYZZ29<F>::add_into(Acc{&d}, Acc{&o});
            };
            XYZZ29<F> acc = XYZZ29<F>::inf();
            acc.zz = FO::zero();
            plus(acc, m3); plus(acc, B.dbl_pt().add(B)); plus(acc, na); plus(acc, nb);
            XYZZ29<F> dd = A, zz_ = A;
            plus(dd, A);                               // doubling branch
            plus(dd, A.dbl_pt().neg_pt());             // -> identity (inverse operands), unless a is the identity
            plus(zz_, XYZZ29<F>::inf());               // identity operand: unchanged
            XYZZ<F> x = r.to_xyzz32(), y = acc.to_xyzz32(), z0 = zz_.to_xyzz32(), a0 = A.to_xyzz32();
            Affine<F> xa = x.to_affine(), ya = y.to_affine(), za = z0.to_affine(), aa = a0.to_affine();
            if (!(xa.x == ya.x) || !(xa.y == ya.y) || !dd.is_inf() || !(za.x == aa.x) || !(za.y == aa.y))
              r = XYZZ29<F>::inf().madd(pa, false);    // poison
          }
        }
        XYZZ29<F> same = A, twice;                    // a + a through add_mem = the doubling branch
        XYZZ29<F>::add_mem(&twice, &same, &A);
        XYZZ29<F> chk = twice.add(A.dbl_pt().neg_pt());
        {
          XYZZ<F> x = r.to_xyzz32(), y = n3.to_xyzz32();
          Affine<F> xa = x.to_affine(), ya = y.to_affine();
          if (!(xa.x == ya.x) || !(xa.y == ya.y) || !(chk.is_inf() || A.is_inf())) r = XYZZ29<F>::inf().madd(pa, false);  // poison
        }
        break;
      }
      case 1: r = A.dbl_pt().madd(pb, false).add(A.neg_pt()); break;
      case 2: r = A.madd(pb, true); break;
      case 3: r = A.dbl_pt(); break;
      default: {   // k * a by double-and-add with k = b's first 8 words: long chains of dbl_pt / add
        const uint32_t* k = (const uint32_t*)&b[i];
        r = XYZZ29<F>::inf();
        for (int bit = 255; bit >= 0; bit--) {
          r = r.dbl_pt();
          if ((k[bit / 32] >> (bit % 32)) & 1) r = (bit & 1) ? r.add(A) : r.madd(pa, false);
        }
        break;
      }
    }
    XYZZ<F> r32 = r.to_xyzz32();
    Jacobian<F> j = r32.to_jacobian();
    o[i] = XYZZ<F>::from_jacobian(j).to_affine();
  }
}
extern "C" int ha_field_op29(int fid, int op, const void* a, const void* b, void* o, size_t n) {
  switch (fid) {
    case 0: field_op29<bn254_fq_params>(op, (const bn_fq*)a, (const bn_fq*)b, (bn_fq*)o, n); break;
    case 1: field_op29<bls12_381_fq_params>(op, (const b381_fq*)a, (const b381_fq*)b, (b381_fq*)o, n); break;
    case 2: field_op29<bls12_377_fq_params>(op, (const b377_fq*)a, (const b377_fq*)b, (b377_fq*)o, n); break;
    case 16: field_op29<bn254_fr_params>(op, (const bn_fr*)a, (const bn_fr*)b, (bn_fr*)o, n); break;
    case 17: field_op29<bls12_381_fr_params>(op, (const b381_fr*)a, (const b381_fr*)b, (b381_fr*)o, n); break;
    case 18: field_op29<bls12_377_fr_params>(op, (const b377_fr*)a, (const b377_fr*)b, (b377_fr*)o, n); break;
    default: return 1;
  }
  return 0;
}
extern "C" int ha_point_op29(int curve, int group, int op, const void* a, const void* b, void* o, size_t n) {
  switch (curve * 2 + group - 1) {
    case 0: point_op29<bn_fq>(op, (const Affine<bn_fq>*)a, (const Affine<bn_fq>*)b, (Affine<bn_fq>*)o, n); break;
    case 1: point_op29<Fp2<bn_fq>>(op, (const Affine<Fp2<bn_fq>>*)a, (const Affine<Fp2<bn_fq>>*)b, (Affine<Fp2<bn_fq>>*)o, n); break;
    case 2: point_op29<b381_fq>(op, (const Affine<b381_fq>*)a, (const Affine<b381_fq>*)b, (Affine<b381_fq>*)o, n); break;
    case 3: point_op29<Fp2<b381_fq>>(op, (const Affine<Fp2<b381_fq>>*)a, (const Affine<Fp2<b381_fq>>*)b, (Affine<Fp2<b381_fq>>*)o, n); break;
    case 4: point_op29<b377_fq>(op, (const Affine<b377_fq>*)a, (const Affine<b377_fq>*)b, (Affine<b377_fq>*)o, n); break;
    case 5: point_op29<Fp2<b377_fq>>(op, (const Affine<Fp2<b377_fq>>*)a, (const Affine<Fp2<b377_fq>>*)b, (Affine<Fp2<b377_fq>>*)o, n); break;
    default: return 1;
  }
  return 0;
}

// ---- the constants fp29.h derives at compile time, dumped for tests/test_constants_independent.py ------------------------
template <class P> static size_t rr_dump(uint32_t* o, size_t cap) {
  using T = RR<P>;
  const size_t need = 4 + 5 * T::N + 2 * (T::NL + 1) + 1;
  if (cap < need) return 0;
  size_t k = 0;
  o[k++] = T::W; o[k++] = T::N; o[k++] = T::SLACK; o[k++] = T::INV;
  for (int i = 0; i < T::N; i++) o[k++] = T::PL.v[i];
  for (int i = 0; i < T::N; i++) o[k++] = T::ONE.v[i];
  for (int i = 0; i < T::N; i++) o[k++] = T::R2.v[i];
  for (int i = 0; i < T::N; i++) o[k++] = T::FROM32.v[i];
  for (int i = 0; i < T::N; i++) o[k++] = T::TO32.v[i];
  for (int i = 0; i <= T::NL; i++) o[k++] = T::R_WORDS.v[i];
  for (int i = 0; i <= T::NL; i++) o[k++] = T::R32SQ_OVER_R_WORDS.v[i];
  o[k++] = T::PTOP;
  return k;
}
extern "C" int ha_rr_consts(int fid, uint32_t* out, size_t cap) {
  switch (fid) {
    case 0: return (int)rr_dump<bn254_fq_params>(out, cap);
    case 1: return (int)rr_dump<bls12_381_fq_params>(out, cap);
    case 2: return (int)rr_dump<bls12_377_fq_params>(out, cap);
    case 16: return (int)rr_dump<bn254_fr_params>(out, cap);
    case 17: return (int)rr_dump<bls12_381_fr_params>(out, cap);
    case 18: return (int)rr_dump<bls12_377_fr_params>(out, cap);
    default: return 0;
  }
}

// ---- the arkworks point codec (csrc/codec_impl.h), host instantiation: BN254, BLS12-381 (zcash form), BLS12-377 ---------------------
#include "../../distributed-groth16_amd/csrc/codec_impl.h"
template <int CURVE> static int codec_run(int group, int decode, int validate, const uint8_t* in, uint8_t* out, size_t n, int* rc) {
  using C = CodecT<CURVE>;
  const size_t fb = C::FB;
  for (size_t i = 0; i < n; i++) {
    if (group == 1) {
      using A = Affine<typename C::Fq>;
      if (decode) { A p = A::inf(); rc[i] = C::decode(in + i * fb, p, validate != 0); memcpy(out + i * sizeof(A), &p, sizeof(A)); }
      else { A p; memcpy(&p, in + i * sizeof(A), sizeof(A)); C::encode(p, out + i * fb); rc[i] = 0; }
    } else {
      using A = Affine<typename C::Fq2>;
      if (decode) { A p = A::inf(); rc[i] = C::decode(in + i * 2 * fb, p, validate != 0); memcpy(out + i * sizeof(A), &p, sizeof(A)); }
      else { A p; memcpy(&p, in + i * sizeof(A), sizeof(A)); C::encode(p, out + i * 2 * fb); rc[i] = 0; }
    }
  }
  return 0;
}
// decode = 0: in = affine Montgomery limbs -> out = compressed bytes; decode = 1: the reverse, rc[i] = the codec's code
extern "C" int ha_codec(int curve, int group, int decode, int validate, const void* in, void* out, size_t n, int* rc) {
  if (curve == 0) return codec_run<0>(group, decode, validate, (const uint8_t*)in, (uint8_t*)out, n, rc);
  if (curve == 1) return codec_run<1>(group, decode, validate, (const uint8_t*)in, (uint8_t*)out, n, rc);
  if (curve == 2) return codec_run<2>(group, decode, validate, (const uint8_t*)in, (uint8_t*)out, n, rc);
  return -1;
}

// glv::split (csrc/glv.h) on n 8-word scalars: h1 / h2 = |k1| / |k2| with the sign in bit 255
extern "C" int ha_glv_split(int curve, const uint32_t* k, uint32_t* h1, uint32_t* h2, size_t n) {
  for (size_t i = 0; i < n; i++) {
    switch (curve) {
      case 0: glv::split<bn254_glv_consts>(k + 8 * i, h1 + 8 * i, h2 + 8 * i); break;
      case 1: glv::split<bls12_381_glv_consts>(k + 8 * i, h1 + 8 * i, h2 + 8 * i); break;
      case 2: glv::split<bls12_377_glv_consts>(k + 8 * i, h1 + 8 * i, h2 + 8 * i); break;
      case 3: glv::split<bn254_g2_glv_consts>(k + 8 * i, h1 + 8 * i, h2 + 8 * i); break;
      default: return -1;
    }
  }
  return 0;
}

// glv::split4 (csrc/glv.h) on n 8-word scalars: h[j] = |k_j| with the sign in bit 255, j < 4 (h: 4 arrays of n x 8 words)
extern "C" int ha_glv_split4(int curve, const uint32_t* k, uint32_t* h0, uint32_t* h1, uint32_t* h2, uint32_t* h3, size_t n) {
  for (size_t i = 0; i < n; i++) {
    switch (curve) {
      case 1: glv::split4<bls12_381_g2_glv4_consts>(k + 8 * i, h0 + 8 * i, h1 + 8 * i, h2 + 8 * i, h3 + 8 * i); break;
      case 2: glv::split4<bls12_377_g2_glv4_consts>(k + 8 * i, h0 + 8 * i, h1 + 8 * i, h2 + 8 * i, h3 + 8 * i); break;
      default: return -1;
    }
  }
  return 0;
}
