// circom `.r1cs` and snarkjs `.zkey` readers -- host side of libdg16 (no GPU work here), the native counterpart of
// the reference's Rust readers: ark-circom/src/circom/r1cs_reader.rs:54-249 (R1CSFile::new) and
// ark-circom/src/zkey.rs:53-388 (read_zkey).  Same acceptance rules and error texts as the reference; the zkey
// reader is zero-copy: points in a zkey are x || y Montgomery limbs with the identity as (0, 0), which is
// libdg16's base layout, so the handle only records where each section lies in the caller's buffer.
#include <stdint.h>
#include <string.h>

#include <map>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/dg16.h"

namespace {

thread_local std::string g_io_error;

int fail(int code, const char* text) {
  g_io_error = text;
  return code;
}

// BN254 scalar / base field moduli, little-endian bytes
const uint8_t kBn254R[32] = {0x01, 0x00, 0x00, 0xf0, 0x93, 0xf5, 0xe1, 0x43, 0x91, 0x70, 0xb9, 0x79, 0x48, 0xe8, 0x33, 0x28,
                             0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45, 0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};
const uint8_t kBn254Q[32] = {0x47, 0xfd, 0x7c, 0xd8, 0x16, 0x8c, 0x20, 0x3c, 0x8d, 0xca, 0x71, 0x68, 0x91, 0x6a, 0x81, 0x97,
                             0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45, 0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};

struct Reader {
  const uint8_t* p;
  size_t n;
  bool has(size_t off, size_t len) const { return off <= n && len <= n - off; }
  uint32_t u32(size_t off) const { uint32_t v; memcpy(&v, p + off, 4); return v; }
  uint64_t u64(size_t off) const { uint64_t v; memcpy(&v, p + off, 8); return v; }
};

struct Section { size_t off, size; };

// iden3 binary container: magic, version, section table {id u32, size u64, payload}
int sections(const Reader& r, std::map<uint32_t, Section>& out, uint32_t& version) {
  if (!r.has(0, 12)) return fail(DG16_ERR_BAD_ARG, "truncated file");
  version = r.u32(4);
  uint32_t count = r.u32(8);
  size_t off = 12;
  for (uint32_t i = 0; i < count; i++) {
    if (!r.has(off, 12)) return fail(DG16_ERR_BAD_ARG, "truncated section table");
    uint32_t id = r.u32(off);
    uint64_t size = r.u64(off + 4);
    off += 12;
    if (!r.has(off, size)) return fail(DG16_ERR_BAD_ARG, "truncated file");
    out.emplace(id, Section{off, (size_t)size});   // the first section of an id wins (zkey.rs:145-147)
    off += size;
  }
  return DG16_OK;
}

struct Csr {
  std::vector<uint32_t> row_ptr, col;
  std::vector<uint8_t> coeff;   // 32 bytes per entry
};

void fill(const Csr& m, dg16_csr* out) {
  out->n_rows = m.row_ptr.empty() ? 0 : m.row_ptr.size() - 1;
  out->nnz = m.col.size();
  out->row_ptr = m.row_ptr.data();
  out->col = m.col.data();
  out->coeff = m.coeff.data();
}

}  // namespace

struct dg16_r1cs {
  dg16_r1cs_header h{};
  Csr m[3];
  std::vector<uint64_t> wire_map;
};

struct dg16_zkey {
  dg16_zkey_header h{};
  const uint8_t* pts[12] = {};
  size_t cnt[12] = {};
  Csr m[2];   // coefficients as stored: value * R^2
};

extern "C" {

const char* dg16_io_error(void) { return g_io_error.c_str(); }

}  // extern "C"

// Nothing may unwind through the C ABI: allocation failures of the containers below (sizes come from file
// fields) are mapped to DG16_ERR_OOM / DG16_ERR_BAD_ARG, and the handle is owned by a unique_ptr until success.
template <class Fn>
static int no_throw(Fn&& fn) {
  try {
    return fn();
  } catch (const std::bad_alloc&) {
    return fail(DG16_ERR_OOM, "out of memory while parsing");
  } catch (const std::length_error&) {
    return fail(DG16_ERR_BAD_ARG, "section size exceeds what can be allocated");
  } catch (const std::exception& e) {
    g_io_error = e.what();
    return DG16_ERR_BAD_ARG;
  } catch (...) {
    return fail(DG16_ERR_BAD_ARG, "unexpected failure while parsing");
  }
}

static int r1cs_parse_impl(const void* data, size_t bytes, dg16_r1cs** out) {
  Reader r{(const uint8_t*)data, bytes};
  if (bytes < 4 || memcmp(r.p, "r1cs", 4) != 0) return fail(DG16_ERR_BAD_ARG, "Invalid magic number");
  std::map<uint32_t, Section> sec;
  uint32_t version = 0;
  if (int rc = sections(r, sec, version)) return rc;
  if (version != 1) return fail(DG16_ERR_UNSUPPORTED, "Unsupported version");
  auto hs = sec.find(1);
  if (hs == sec.end()) return fail(DG16_ERR_BAD_ARG, "No section offset for header type found");
  const size_t ho = hs->second.off;
  if (hs->second.size < 4) return fail(DG16_ERR_BAD_ARG, "Invalid header section size");
  const uint32_t field_size = r.u32(ho);
  if (field_size != 32) return fail(DG16_ERR_UNSUPPORTED, "This parser only supports 32-byte fields");
  if (hs->second.size != 32 + field_size) return fail(DG16_ERR_BAD_ARG, "Invalid header section size");
  if (memcmp(r.p + ho + 4, kBn254R, 32) != 0) return fail(DG16_ERR_UNSUPPORTED, "This parser only supports bn256");
  std::unique_ptr<dg16_r1cs> f(new dg16_r1cs());
  f->h.n_wires = r.u32(ho + 36);
  f->h.n_pub_out = r.u32(ho + 40);
  f->h.n_pub_in = r.u32(ho + 44);
  f->h.n_prv_in = r.u32(ho + 48);
  f->h.n_labels = r.u64(ho + 52);
  f->h.n_constraints = r.u32(ho + 60);
  auto cs = sec.find(2);
  if (cs == sec.end()) return fail(DG16_ERR_BAD_ARG, "No section offset for constraint type found");
  size_t p = cs->second.off;
  const size_t end = cs->second.off + cs->second.size;
  for (int k = 0; k < 3; k++) f->m[k].row_ptr.assign(1, 0);
  for (uint32_t c = 0; c < f->h.n_constraints; c++)
    for (int k = 0; k < 3; k++) {
      if (p + 4 > end) return fail(DG16_ERR_BAD_ARG, "truncated constraint section");
      const uint32_t n_vec = r.u32(p);
      p += 4;
      if ((size_t)n_vec * 36 > end - p) return fail(DG16_ERR_BAD_ARG, "truncated constraint section");
      Csr& m = f->m[k];
      for (uint32_t j = 0; j < n_vec; j++) {
        // a wire id indexes the assignment downstream (dg16_qap): the reference's Rust indexing would panic on
        // an id >= n_wires, here the file is rejected
        if (r.u32(p) >= f->h.n_wires) return fail(DG16_ERR_BAD_ARG, "coefficient out of range (wire >= n_wires)");
        m.col.push_back(r.u32(p));
        m.coeff.insert(m.coeff.end(), r.p + p + 4, r.p + p + 36);
        p += 36;
      }
      m.row_ptr.push_back((uint32_t)m.col.size());
    }
  auto ms = sec.find(3);
  if (ms != sec.end()) {
    if (ms->second.size != (size_t)f->h.n_wires * 8) return fail(DG16_ERR_BAD_ARG, "Invalid map section size");
    f->wire_map.resize(f->h.n_wires);
    if (f->h.n_wires) memcpy(f->wire_map.data(), r.p + ms->second.off, (size_t)f->h.n_wires * 8);
    if (f->h.n_wires && f->wire_map[0] != 0) return fail(DG16_ERR_BAD_ARG, "Wire 0 should always be mapped to 0");
    f->h.has_wire_map = 1;
  }
  *out = f.release();
  return DG16_OK;
}

extern "C" {

int dg16_r1cs_parse(const void* data, size_t bytes, dg16_r1cs** out) {
  if (!data || !out) return fail(DG16_ERR_BAD_ARG, "null argument");
  *out = nullptr;
  return no_throw([&] { return r1cs_parse_impl(data, bytes, out); });
}

int dg16_r1cs_header_get(const dg16_r1cs* f, dg16_r1cs_header* out) {
  if (!f || !out) return fail(DG16_ERR_BAD_ARG, "null argument");
  *out = f->h;
  return DG16_OK;
}

int dg16_r1cs_matrix(const dg16_r1cs* f, int which, dg16_csr* out) {
  if (!f || !out || which < 0 || which > 2) return fail(DG16_ERR_BAD_ARG, "matrix must be 0 (A), 1 (B) or 2 (C)");
  fill(f->m[which], out);
  return DG16_OK;
}

int dg16_r1cs_wire_map(const dg16_r1cs* f, const uint64_t** map) {
  if (!f || !map) return fail(DG16_ERR_BAD_ARG, "null argument");
  *map = f->h.has_wire_map ? f->wire_map.data() : nullptr;
  return DG16_OK;
}

void dg16_r1cs_free(dg16_r1cs* f) { delete f; }

}  // extern "C"

static int zkey_parse_impl(const void* data, size_t bytes, dg16_zkey** out) {
  Reader r{(const uint8_t*)data, bytes};
  if (bytes < 4 || memcmp(r.p, "zkey", 4) != 0) return fail(DG16_ERR_BAD_ARG, "Invalid magic number");
  std::map<uint32_t, Section> sec;
  uint32_t version = 0;
  if (int rc = sections(r, sec, version)) return rc;
  for (uint32_t id = 1; id <= 9; id++)
    if (!sec.count(id)) {
      g_io_error = "missing section " + std::to_string(id);
      return DG16_ERR_BAD_ARG;
    }
  if (sec[1].size < 4 || r.u32(sec[1].off) != 1) return fail(DG16_ERR_UNSUPPORTED, "not a Groth16 key");
  // ---- section 2: header (zkey.rs:296-330) ----
  const Section hs = sec[2];
  const size_t need = 4 + 32 + 4 + 32 + 12 + 3 * 64 + 3 * 128;
  if (hs.size < need) return fail(DG16_ERR_BAD_ARG, "header section too short");
  size_t p = hs.off;
  if (r.u32(p) != 32 || memcmp(r.p + p + 4, kBn254Q, 32) != 0) return fail(DG16_ERR_UNSUPPORTED, "base field is not BN254's");
  p += 36;
  if (r.u32(p) != 32 || memcmp(r.p + p + 4, kBn254R, 32) != 0) return fail(DG16_ERR_UNSUPPORTED, "scalar field is not BN254's");
  p += 36;
  std::unique_ptr<dg16_zkey> z(new dg16_zkey());
  z->h.n_vars = r.u32(p);
  z->h.n_public = r.u32(p + 4);
  z->h.domain_size = r.u32(p + 8);
  p += 12;
  if (z->h.domain_size == 0 || (z->h.domain_size & (z->h.domain_size - 1)))
    return fail(DG16_ERR_BAD_ARG, "domain size is not a power of two");
  if (z->h.n_vars < z->h.n_public + 1) return fail(DG16_ERR_BAD_ARG, "n_vars < n_public + 1");
  // alpha_g1, beta_g1, beta_g2, gamma_g2, delta_g1, delta_g2 (zkey.rs:259-276)
  const size_t fixed_bytes[6] = {64, 64, 128, 128, 64, 128};
  for (int i = 0; i < 6; i++) {
    z->pts[i] = r.p + p;
    z->cnt[i] = 1;
    p += fixed_bytes[i];
  }
  const uint32_t nv = z->h.n_vars, np = z->h.n_public;
  const struct { int which; uint32_t id; size_t count, bytes; } qs[6] = {
      {DG16_ZKEY_IC, 3, (size_t)np + 1, 64},      {DG16_ZKEY_A, 5, nv, 64}, {DG16_ZKEY_B1, 6, nv, 64},
      {DG16_ZKEY_B2, 7, nv, 128},                 {DG16_ZKEY_L, 8, (size_t)nv - np - 1, 64},
      {DG16_ZKEY_H, 9, z->h.domain_size, 64}};
  for (const auto& q : qs) {
    const Section s = sec[q.id];
    if (s.size < q.count * q.bytes) {
      g_io_error = "section " + std::to_string(q.id) + " too short";
      return DG16_ERR_BAD_ARG;
    }
    z->pts[q.which] = r.p + s.off;
    z->cnt[q.which] = q.count;
  }
  // ---- section 4: coefficients (zkey.rs:149-198) ----
  const Section cs = sec[4];
  if (cs.size < 4) return fail(DG16_ERR_BAD_ARG, "coefficient section too short");
  const uint32_t n_coeffs = r.u32(cs.off);
  if (cs.size < 4 + (size_t)n_coeffs * 44) return fail(DG16_ERR_BAD_ARG, "coefficient section too short");
  uint32_t max_row = 0;
  for (uint32_t i = 0; i < n_coeffs; i++) {
    const size_t o = cs.off + 4 + (size_t)i * 44;
    const uint32_t matrix = r.u32(o), row = r.u32(o + 4);
    // `signal` indexes the assignment downstream (dg16_qap): reject what the reference's indexing would panic on
    if (matrix > 1 || row >= z->h.domain_size || r.u32(o + 8) >= z->h.n_vars)
      return fail(DG16_ERR_BAD_ARG, "coefficient out of range");
    if (row > max_row) max_row = row;
  }
  // the rows above num_constraints (the public-input rows snarkjs appends) are dropped (zkey.rs:176-180)
  z->h.num_constraints = max_row >= np ? max_row - np : 0;
  const uint32_t nc = z->h.num_constraints;
  for (int k = 0; k < 2; k++) z->m[k].row_ptr.assign((size_t)nc + 1, 0);
  for (uint32_t i = 0; i < n_coeffs; i++) {
    const size_t o = cs.off + 4 + (size_t)i * 44;
    const uint32_t row = r.u32(o + 4);
    if (row < nc) z->m[r.u32(o)].row_ptr[row + 1]++;
  }
  std::vector<uint32_t> cursor[2];
  for (int k = 0; k < 2; k++) {
    Csr& m = z->m[k];
    for (uint32_t i = 0; i < nc; i++) m.row_ptr[i + 1] += m.row_ptr[i];
    m.col.resize(m.row_ptr[nc]);
    m.coeff.resize((size_t)m.row_ptr[nc] * 32);
    cursor[k].assign(m.row_ptr.begin(), m.row_ptr.end() - 1);
  }
  for (uint32_t i = 0; i < n_coeffs; i++) {       // file order inside a row, like the reference's push() loop
    const size_t o = cs.off + 4 + (size_t)i * 44;
    const uint32_t k = r.u32(o), row = r.u32(o + 4);
    if (row >= nc) continue;
    const uint32_t at = cursor[k][row]++;
    z->m[k].col[at] = r.u32(o + 8);
    memcpy(z->m[k].coeff.data() + (size_t)at * 32, r.p + o + 12, 32);
  }
  *out = z.release();
  return DG16_OK;
}

extern "C" {

int dg16_zkey_parse(const void* data, size_t bytes, dg16_zkey** out) {
  if (!data || !out) return fail(DG16_ERR_BAD_ARG, "null argument");
  *out = nullptr;
  return no_throw([&] { return zkey_parse_impl(data, bytes, out); });
}

int dg16_zkey_header_get(const dg16_zkey* z, dg16_zkey_header* out) {
  if (!z || !out) return fail(DG16_ERR_BAD_ARG, "null argument");
  *out = z->h;
  return DG16_OK;
}

int dg16_zkey_points(const dg16_zkey* z, int which, const void** ptr, size_t* count) {
  if (!z || !ptr || !count || which < 0 || which > DG16_ZKEY_H) return fail(DG16_ERR_BAD_ARG, "unknown zkey section");
  *ptr = z->pts[which];
  *count = z->cnt[which];
  return DG16_OK;
}

int dg16_zkey_matrix(const dg16_zkey* z, int which, dg16_csr* out) {
  if (!z || !out || which < 0 || which > 1) return fail(DG16_ERR_BAD_ARG, "matrix must be 0 (A) or 1 (B)");
  fill(z->m[which], out);
  return DG16_OK;
}

void dg16_zkey_free(dg16_zkey* z) { delete z; }

}  // extern "C"
