// sunscreen_amd/csrc/wire.cpp -- the SEAL 4.0 binary wire format for the objects that cross the evaluator
// boundary: Ciphertext, Plaintext, KSwitchKeys (RelinKeys / GaloisKeys).
//
// Reference call sites: Ciphertext::{as_bytes,from_bytes} (seal_fhe/src/plaintext_ciphertext.rs:451-497 ->
// Ciphertext_SaveSize / Ciphertext_Save / Ciphertext_Load), KSwitchKeys_Save/Load
// (seal_fhe/src/key_generator.rs:493-573, 649-729), Plaintext_Save/Load (plaintext_ciphertext.rs:100-160).
//
// Layout (decoded from, and tested against, the reference's fixtures seal_fhe/tests/data/*.bin):
//   16-byte header  { u16 magic 0xA15E, u8 header_size 16, u8 major 4, u8 minor 0, u8 compr_mode (0 none, 2 zstd),
//                     u16 reserved 0, u64 total_size }
//   body (zstd-compressed as ONE frame when compr_mode = 2):
//     Ciphertext : parms_id[32], u8 is_ntt_form, u64 size, u64 poly_modulus_degree, u64 coeff_modulus_size,
//                  f64 scale, u64 correction_factor, DynArray<u64>
//     Plaintext  : parms_id[32], u64 coeff_count, f64 scale, DynArray<u64>
//     KSwitchKeys: parms_id[32], u64 dim1, then per index: u64 dim2, then dim2 x PublicKey objects, each a complete
//                  uncompressed SEAL object (header + Ciphertext members, is_ntt_form = 1, size 2, K+1 residues)
//     DynArray   : its own 16-byte header (compr_mode 0), u64 count, count x u64
//   parms_id = BLAKE2b-256 over the little-endian u64 words [scheme = 1 (BFV), n, q_0 .. q_{k-1}, t]
//             (verified on the fixtures: the key-level parms_id of n=8192, create(8192,[50,30,30,50,50]),
//              t = batching(8192,20) is 128ba7b6...06aae084).
// zstd is loaded at run time from libzstd.so.1 (no headers in this image); without it only compr_mode 0 works.
#include "wire.hpp"

#include <dlfcn.h>

#include <cstring>
#include <mutex>

namespace hipbfv {

// ------------------------------------------------------------------ BLAKE2b (RFC 7693), unkeyed
namespace {
const unsigned long long kIv[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                         0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
const uint8_t kSigma[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};

inline unsigned long long rotr(unsigned long long x, int n) { return (x >> n) | (x << (64 - n)); }

void compress(unsigned long long h[8], const uint8_t block[128], unsigned long long t, bool last) {
  unsigned long long m[16], v[16];
  std::memcpy(m, block, 128);
  for (int i = 0; i < 8; i++) v[i] = h[i], v[i + 8] = kIv[i];
  v[12] ^= t;
  if (last) v[14] = ~v[14];
  auto G = [&](int a, int b, int c, int d, unsigned long long x, unsigned long long y) {
    v[a] = v[a] + v[b] + x;
    v[d] = rotr(v[d] ^ v[a], 32);
    v[c] = v[c] + v[d];
    v[b] = rotr(v[b] ^ v[c], 24);
    v[a] = v[a] + v[b] + y;
    v[d] = rotr(v[d] ^ v[a], 16);
    v[c] = v[c] + v[d];
    v[b] = rotr(v[b] ^ v[c], 63);
  };
  for (int r = 0; r < 12; r++) {
    const uint8_t* s = kSigma[r];
    G(0, 4, 8, 12, m[s[0]], m[s[1]]);
    G(1, 5, 9, 13, m[s[2]], m[s[3]]);
    G(2, 6, 10, 14, m[s[4]], m[s[5]]);
    G(3, 7, 11, 15, m[s[6]], m[s[7]]);
    G(0, 5, 10, 15, m[s[8]], m[s[9]]);
    G(1, 6, 11, 12, m[s[10]], m[s[11]]);
    G(2, 7, 8, 13, m[s[12]], m[s[13]]);
    G(3, 4, 9, 14, m[s[14]], m[s[15]]);
  }
  for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}
}  // namespace

void blake2b_256(const void* data, size_t len, uint8_t out[32]) {
  unsigned long long h[8];
  for (int i = 0; i < 8; i++) h[i] = kIv[i];
  h[0] ^= 0x01010000ull ^ 32ull;  // digest length 32, no key, fanout 1, depth 1
  const uint8_t* p = static_cast<const uint8_t*>(data);
  unsigned long long t = 0;
  uint8_t block[128];
  while (len > 128) {
    t += 128;
    compress(h, p, t, false);
    p += 128;
    len -= 128;
  }
  std::memset(block, 0, 128);
  std::memcpy(block, p, len);
  t += len;
  compress(h, block, t, true);
  std::memcpy(out, h, 32);
}

void seal_parms_id(unsigned long long n, const unsigned long long* primes, size_t count, unsigned long long t, uint8_t out[32]) {
  std::vector<unsigned long long> words;
  words.push_back(1);  // scheme_type::bfv
  words.push_back(n);
  for (size_t i = 0; i < count; i++) words.push_back(primes[i]);
  if (t) words.push_back(t);
  blake2b_256(words.data(), words.size() * 8, out);
}

// ------------------------------------------------------------------ zstd through dlopen
namespace {
struct Zstd {
  void* lib = nullptr;
  size_t (*compressBound)(size_t) = nullptr;
  size_t (*compress)(void*, size_t, const void*, size_t, int) = nullptr;
  size_t (*decompress)(void*, size_t, const void*, size_t) = nullptr;
  unsigned long long (*frameContentSize)(const void*, size_t) = nullptr;
  unsigned (*isError)(size_t) = nullptr;
  bool ok() const { return lib && compressBound && compress && decompress && frameContentSize && isError; }
};
Zstd& zstd() {
  static Zstd z;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"libzstd.so.1", "libzstd.so"}) {
      z.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (z.lib) break;
    }
    if (!z.lib) return;
    z.compressBound = (size_t(*)(size_t))dlsym(z.lib, "ZSTD_compressBound");
    z.compress = (size_t(*)(void*, size_t, const void*, size_t, int))dlsym(z.lib, "ZSTD_compress");
    z.decompress = (size_t(*)(void*, size_t, const void*, size_t))dlsym(z.lib, "ZSTD_decompress");
    z.frameContentSize = (unsigned long long (*)(const void*, size_t))dlsym(z.lib, "ZSTD_getFrameContentSize");
    z.isError = (unsigned (*)(size_t))dlsym(z.lib, "ZSTD_isError");
  });
  return z;
}

struct Writer {
  std::vector<uint8_t>& b;
  void raw(const void* p, size_t n) {
    const uint8_t* q = static_cast<const uint8_t*>(p);
    b.insert(b.end(), q, q + n);
  }
  void u8(uint8_t v) { b.push_back(v); }
  void u64(unsigned long long v) { raw(&v, 8); }
  void f64(double v) { raw(&v, 8); }
  void header(uint8_t compr, unsigned long long total) {
    const uint8_t h[8] = {0x5E, 0xA1, 16, 4, 0, compr, 0, 0};
    raw(h, 8);
    u64(total);
  }
  void dynarray(const unsigned long long* data, size_t count) {
    header(0, 16 + 8 + 8 * (unsigned long long)count);
    u64(count);
    raw(data, 8 * count);
  }
};

struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  bool raw(void* dst, size_t n) {
    if ((size_t)(end - p) < n) return ok = false;
    std::memcpy(dst, p, n);
    p += n;
    return true;
  }
  uint8_t u8() {
    uint8_t v = 0;
    raw(&v, 1);
    return v;
  }
  unsigned long long u64() {
    unsigned long long v = 0;
    raw(&v, 8);
    return v;
  }
  double f64() {
    double v = 0;
    raw(&v, 8);
    return v;
  }
  // returns compr mode (or -1) and the object's total size
  int header(unsigned long long* total) {
    uint8_t h[8];
    if (!raw(h, 8)) return -1;
    *total = u64();
    if (h[0] != 0x5E || h[1] != 0xA1 || h[2] != 16 || h[3] != 4) {
      ok = false;
      return -1;
    }
    return h[5];
  }
  bool dynarray(std::vector<unsigned long long>* out, size_t max_count) {
    unsigned long long total = 0;
    if (header(&total) != 0) return ok = false;
    const unsigned long long count = u64();
    if (!ok || count > max_count || total != 16 + 8 + 8 * count) return ok = false;
    out->resize(count);
    return raw(out->data(), 8 * count);
  }
};

// wrap `body` into a SEAL object (compress if requested)
int finish(const std::vector<uint8_t>& body, int compr, std::vector<uint8_t>* out) {
  out->clear();
  Writer w{*out};
  if (compr == 0) {
    w.header(0, 16 + body.size());
    w.raw(body.data(), body.size());
    return kWireOk;
  }
  if (compr != 2) return kWireBadArg;  // zlib is not produced by seal_fhe (CompressionType::ZStd everywhere)
  Zstd& z = zstd();
  if (!z.ok()) return kWireNoZstd;
  std::vector<uint8_t> tmp(z.compressBound(body.size()));
  const size_t got = z.compress(tmp.data(), tmp.size(), body.data(), body.size(), 3);
  if (z.isError(got)) return kWireIo;
  w.header(2, 16 + got);
  w.raw(tmp.data(), got);
  return kWireOk;
}

// strip the outer header, decompress: body bytes + number of input bytes consumed
// `max_body` bounds the decompressed size by what the caller's context can legally hold (a tiny zstd frame may declare
// gigabytes: the bytes are untrusted); 0 = the format-wide ceiling
int open_object(const uint8_t* in, size_t size, std::vector<uint8_t>* body, size_t* consumed, size_t max_body) {
  if (!max_body) max_body = (size_t)1 << 32;
  Reader r{in, in + size};
  unsigned long long total = 0;
  const int compr = r.header(&total);
  if (compr < 0 || total < 16 || total > size) return kWireIo;
  *consumed = total;
  const uint8_t* payload = in + 16;
  const size_t plen = total - 16;
  if (compr == 0) {
    body->assign(payload, payload + plen);
    return kWireOk;
  }
  if (compr != 2) return kWireBadArg;
  Zstd& z = zstd();
  if (!z.ok()) return kWireNoZstd;
  unsigned long long raw = z.frameContentSize(payload, plen);
  if (raw == ~0ull || raw == ~0ull - 1 || raw > max_body) return kWireIo;
  body->resize(raw);
  const size_t got = z.decompress(body->data(), body->size(), payload, plen);
  if (z.isError(got) || got != raw) return kWireIo;
  return kWireOk;
}
}  // namespace

bool wire_zstd_available() { return zstd().ok(); }

int wire_pack_ciphertext(const uint8_t parms_id[32], bool is_ntt, unsigned long long size, unsigned long long n, unsigned long long k, const unsigned long long* data, int compr,
                         std::vector<uint8_t>* out) {
  std::vector<uint8_t> body;
  Writer w{body};
  w.raw(parms_id, 32);
  w.u8(is_ntt ? 1 : 0);
  w.u64(size);
  w.u64(n);
  w.u64(k);
  w.f64(1.0);
  w.u64(1);
  w.dynarray(data, size * n * k);
  return finish(body, compr, out);
}

int wire_unpack_ciphertext(const uint8_t* in, size_t in_size, WireCiphertext* ct, size_t* consumed, size_t max_body) {
  std::vector<uint8_t> body;
  if (int rc = open_object(in, in_size, &body, consumed, max_body)) return rc;
  Reader r{body.data(), body.data() + body.size()};
  r.raw(ct->parms_id, 32);
  ct->is_ntt = r.u8() != 0;
  ct->size = r.u64();
  ct->n = r.u64();
  ct->k = r.u64();
  ct->scale = r.f64();
  ct->correction = r.u64();
  if (!r.ok || ct->size > 64 || ct->n > (1u << 20) || ct->k > 64) return kWireIo;
  if (!r.dynarray(&ct->data, (size_t)1 << 28)) return kWireIo;
  // Ciphertext::save_members with a seed marker: half the data, then a UniformRandomGeneratorInfo object (header, u8 type, 64-byte seed)
  if (ct->size == 2 && ct->data.size() == ct->n * ct->k && (size_t)(r.end - r.p) >= 16 + 1 + 64) return kWireSeeded;
  if (ct->data.size() != ct->size * ct->n * ct->k) return kWireIo;
  return kWireOk;
}

int wire_pack_plaintext(const uint8_t parms_id[32], const unsigned long long* coeffs, unsigned long long count, int compr, std::vector<uint8_t>* out) {
  std::vector<uint8_t> body;
  Writer w{body};
  w.raw(parms_id, 32);
  w.u64(count);
  w.f64(1.0);
  w.dynarray(coeffs, count);
  return finish(body, compr, out);
}

int wire_unpack_plaintext(const uint8_t* in, size_t in_size, WirePlaintext* pt, size_t* consumed, size_t max_body) {
  std::vector<uint8_t> body;
  if (int rc = open_object(in, in_size, &body, consumed, max_body)) return rc;
  Reader r{body.data(), body.data() + body.size()};
  r.raw(pt->parms_id, 32);
  const unsigned long long count = r.u64();
  pt->scale = r.f64();
  if (!r.ok || !r.dynarray(&pt->coeffs, (size_t)1 << 28) || pt->coeffs.size() != count) return kWireIo;
  return kWireOk;
}

int wire_pack_kswitch(const uint8_t parms_id[32], unsigned long long n, unsigned long long kk, const std::vector<std::vector<const unsigned long long*>>& keys, int compr,
                      std::vector<uint8_t>* out) {
  std::vector<uint8_t> body;
  Writer w{body};
  w.raw(parms_id, 32);
  w.u64(keys.size());
  for (const auto& entry : keys) {
    w.u64(entry.size());
    for (const unsigned long long* pk : entry) {
      std::vector<uint8_t> obj;
      if (int rc = wire_pack_ciphertext(parms_id, true, 2, n, kk, pk, 0, &obj)) return rc;
      w.raw(obj.data(), obj.size());
    }
  }
  return finish(body, compr, out);
}

int wire_unpack_kswitch(const uint8_t* in, size_t in_size, WireKSwitchKeys* ks, size_t* consumed, size_t max_body, size_t max_index) {
  std::vector<uint8_t> body;
  if (int rc = open_object(in, in_size, &body, consumed, max_body)) return rc;
  Reader r{body.data(), body.data() + body.size()};
  r.raw(ks->parms_id, 32);
  const unsigned long long dim1 = r.u64();
  // an index is (galois_elt - 1) / 2 < N: the context bounds it; every entry costs at least its 8-byte length word
  if (!r.ok || dim1 > (max_index ? max_index : (1u << 17)) || dim1 > body.size() / 8) return kWireIo;
  ks->keys.resize(dim1);
  for (unsigned long long i = 0; i < dim1; i++) {
    const unsigned long long dim2 = r.u64();
    if (!r.ok || dim2 > 64) return kWireIo;
    ks->keys[i].resize(dim2);
    for (unsigned long long j = 0; j < dim2; j++) {
      size_t used = 0;
      if (int rc = wire_unpack_ciphertext(r.p, (size_t)(r.end - r.p), &ks->keys[i][j], &used)) return rc;
      r.p += used;
    }
  }
  return kWireOk;
}

}  // namespace hipbfv
