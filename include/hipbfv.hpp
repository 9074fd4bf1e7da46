// include/hipbfv.hpp -- header-only C++17 host mirror of the `seal_fhe` crate's surface over the C ABI of hipbfv.h.
//
// The reference binds SEAL's C API from Rust (`seal_fhe/src/*.rs`); no Rust toolchain exists in this image, so the compiled-
// language host side of the boundary is this header: the same type and method names, argument meaning, ownership and error
// behaviour as the crate, one C-ABI call per method -- what a maintainer would otherwise write in `seal_fhe/src` against
// bindgen (INTEGRATION.md shows that binding).  Python mirrors the same surface in sunscreen_amd/seal.py for the tests.
//
//   crate item (file:line)                                            here
//   Error / Result<T>            (error.rs:5-91)                       seal_fhe::Error (exception) carrying the same variants
//   Modulus, CoeffModulus, PlainModulus, SecurityLevel (modulus.rs)    same names
//   BfvEncryptionParametersBuilder, EncryptionParameters               same names (encryption_parameters.rs:196-300)
//   Context::new / new_insecure  (context.rs:63-103)                   Context
//   Plaintext, Ciphertext        (plaintext_ciphertext.rs)             same names; as_bytes / from_bytes = SEAL wire format
//   KeyGenerator, SecretKey, PublicKey, RelinearizationKeys, GaloisKeys (key_generator.rs)
//   BFVEncoder                   (encoder.rs:40-215)                   BFVEncoder
//   Encryptor / Decryptor        (encryptor_decryptor.rs)              Encryptor (public-key, symmetric), Decryptor
//   trait Evaluator + BFVEvaluator (evaluator.rs:7-280, bfv_evaluator.rs:27-244)   BFVEvaluator, all 30 methods
//
// Ownership: every wrapper owns exactly one opaque handle and destroys it in its destructor (the crate's Drop); copies are
// deep (X_Create2 / Create5, the crate's Clone); moves transfer the handle.  All calls are synchronous and thread-safe on a
// shared BFVEvaluator (sunscreen_runtime/src/run.rs:415-469).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "hipbfv.h"

// The namespace is the crate's name, NOT `hipbfv`: libhipbfv.so exports its internal C++ symbols (namespace hipbfv, e.g.
// hipbfv::Context), and an inline function of the same qualified name defined here would interpose them at -O0.
namespace seal_fhe {

// seal_fhe::Error (error.rs:5-62) with the HRESULT mapping of `impl From<c_long> for Error` (error.rs:65-78)
class Error : public std::runtime_error {
 public:
  enum Kind { InvalidArgument, InvalidPointer, OutOfMemory, Unexpected, InternalError, Unknown, SerializationError };
  Error(Kind kind, long hresult, const std::string& what) : std::runtime_error(what), kind_(kind), hresult_(hresult) {}
  Kind kind() const { return kind_; }
  long hresult() const { return hresult_; }
  static Kind kind_of(long hr) {
    if (hr == HIPBFV_E_POINTER) return InvalidPointer;
    if (hr == HIPBFV_E_INVALIDARG) return InvalidArgument;
    if (hr == HIPBFV_E_OUTOFMEMORY) return OutOfMemory;
    if (hr == HIPBFV_E_UNEXPECTED) return Unexpected;
    if (hr == HIPBFV_COR_E_IO || hr == HIPBFV_COR_E_INVALIDOPERATION) return InternalError;
    return Unknown;
  }

 private:
  Kind kind_;
  long hresult_;
};

// convert_seal_error (error.rs:85-91)
inline void check(long hr) {
  if (hr == HIPBFV_S_OK) return;
  char msg[256] = {0};
  (void)hipbfv_last_error(msg, sizeof msg);
  static const char* names[] = {"InvalidArgument", "InvalidPointer", "OutOfMemory", "Unexpected", "InternalError", "Unknown", "SerializationError"};
  const Error::Kind k = Error::kind_of(hr);
  throw Error(k, hr, std::string(names[k]) + (msg[0] ? std::string(": ") + msg : std::string()));
}

namespace detail {
// one owned opaque handle; D = X_Destroy, C = deep-copy constructor (X_Create2 / Create5) or nullptr
template <long (*D)(void*), long (*C)(void*, void**)>
class Handle {
 public:
  Handle() = default;
  explicit Handle(void* h) : h_(h) {}
  Handle(const Handle& o) {
    if (o.h_) check(C(o.h_, &h_));
  }
  Handle(Handle&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
  Handle& operator=(Handle o) noexcept {
    std::swap(h_, o.h_);
    return *this;
  }
  ~Handle() {
    if (h_) (void)D(h_);
  }
  void* get_handle() const { return h_; }  // the crate's get_handle(): borrowed, must not be destroyed

 protected:
  void* h_ = nullptr;
};
inline long no_copy(void*, void**) { return HIPBFV_E_INVALIDARG; }
}  // namespace detail

// ---- modulus.rs ----
enum class SecurityLevel : int { TC128 = 128, TC192 = 192, TC256 = 256 };

class Modulus : public detail::Handle<Modulus_Destroy, Modulus_Create2> {
 public:
  using Handle::Handle;
  explicit Modulus(uint64_t value) { check(Modulus_Create1(value, &h_)); }  // Modulus::new (modulus.rs:104-113)
  uint64_t value() const {
    uint64_t v = 0;
    check(Modulus_Value(h_, &v));
    return v;
  }
};

struct CoeffModulus {
  // CoeffModulus::create (modulus.rs:174-194): primes of the given bit sizes, == 1 (mod 2 * degree)
  static std::vector<Modulus> create(uint64_t degree, const std::vector<int>& bit_sizes) {
    std::vector<void*> out(bit_sizes.size(), nullptr);
    std::vector<int> bits(bit_sizes);
    check(CoeffModulus_Create1(degree, bits.size(), bits.data(), out.data()));
    std::vector<Modulus> r;
    for (void* h : out) r.emplace_back(h);
    return r;
  }
  // CoeffModulus::bfv_default (modulus.rs:203-229)
  static std::vector<Modulus> bfv_default(uint64_t degree, SecurityLevel level = SecurityLevel::TC128) {
    uint64_t len = 0;
    check(CoeffModulus_BFVDefault(degree, (int)level, &len, nullptr));
    std::vector<void*> out(len, nullptr);
    check(CoeffModulus_BFVDefault(degree, (int)level, &len, out.data()));
    std::vector<Modulus> r;
    for (void* h : out) r.emplace_back(h);
    return r;
  }
  static uint32_t max_bit_count(uint64_t degree, SecurityLevel level = SecurityLevel::TC128) {  // modulus.rs:236-246
    int bits = 0;
    check(CoeffModulus_MaxBitCount(degree, (int)level, &bits));
    return (uint32_t)bits;
  }
};

struct PlainModulus {
  // PlainModulus::batching (modulus.rs:257-263): a prime of bit_size bits supporting batching = CoeffModulus::create(.., [bits])[0]
  static Modulus batching(uint64_t degree, uint32_t bit_size) { return std::move(CoeffModulus::create(degree, {(int)bit_size})[0]); }
  static Modulus raw(uint64_t value) { return Modulus(value); }  // modulus.rs:269-271
};

// ---- encryption_parameters.rs ----
enum class SchemeType : uint8_t { None = 0, Bfv = 1 };

class EncryptionParameters : public detail::Handle<EncParams_Destroy, detail::no_copy> {
 public:
  using Handle::Handle;
  EncryptionParameters(const EncryptionParameters&) = delete;
  EncryptionParameters(EncryptionParameters&&) = default;
  uint64_t get_poly_modulus_degree() const {
    uint64_t d = 0;
    check(EncParams_GetPolyModulusDegree(h_, &d));
    return d;
  }
  SchemeType get_scheme() const {
    uint8_t s = 0;
    check(EncParams_GetScheme(h_, &s));
    return (SchemeType)s;
  }
  // libhipbfv returns NEW Modulus objects from both getters (the crate clones SEAL's borrowed ones and forgets them,
  // encryption_parameters.rs:125-190; the same objects are simply adopted here, as sunscreen_amd/seal.py does)
  Modulus get_plain_modulus() const {
    void* h = nullptr;
    check(EncParams_GetPlainModulus(h_, &h));
    return Modulus(h);
  }
  std::vector<Modulus> get_coefficient_modulus() const {
    uint64_t len = 0;
    check(EncParams_GetCoeffModulus(h_, &len, nullptr));
    std::vector<void*> hs(len, nullptr);
    check(EncParams_GetCoeffModulus(h_, &len, hs.data()));
    std::vector<Modulus> r;
    for (void* h : hs) r.emplace_back(h);
    return r;
  }
};

// BfvEncryptionParametersBuilder (encryption_parameters.rs:196-300)
class BfvEncryptionParametersBuilder {
 public:
  BfvEncryptionParametersBuilder& set_poly_modulus_degree(uint64_t degree) {
    degree_ = degree, have_degree_ = true;
    return *this;
  }
  BfvEncryptionParametersBuilder& set_coefficient_modulus(std::vector<Modulus> m) {
    coeff_ = std::move(m);
    return *this;
  }
  BfvEncryptionParametersBuilder& set_plain_modulus_u64(uint64_t t) {
    plain_u64_ = t, have_plain_ = 1;
    return *this;
  }
  BfvEncryptionParametersBuilder& set_plain_modulus(Modulus t) {
    plain_ = std::move(t), have_plain_ = 2;
    return *this;
  }
  EncryptionParameters build() const {
    // the crate returns Error::DegreeNotSet / CoefficientModulusNotSet / PlainModulusNotSet (encryption_parameters.rs:270-300)
    if (!have_degree_) throw Error(Error::InvalidArgument, HIPBFV_E_INVALIDARG, "DegreeNotSet");
    if (coeff_.empty()) throw Error(Error::InvalidArgument, HIPBFV_E_INVALIDARG, "CoefficientModulusNotSet");
    if (!have_plain_) throw Error(Error::InvalidArgument, HIPBFV_E_INVALIDARG, "PlainModulusNotSet");
    void* h = nullptr;
    check(EncParams_Create1((uint8_t)SchemeType::Bfv, &h));
    EncryptionParameters p(h);
    check(EncParams_SetPolyModulusDegree(h, degree_));
    std::vector<void*> hs;
    for (const Modulus& m : coeff_) hs.push_back(m.get_handle());
    check(EncParams_SetCoeffModulus(h, hs.size(), hs.data()));
    if (have_plain_ == 2)
      check(EncParams_SetPlainModulus1(h, plain_.get_handle()));
    else
      check(EncParams_SetPlainModulus2(h, plain_u64_));
    return p;
  }

 private:
  uint64_t degree_ = 0, plain_u64_ = 0;
  bool have_degree_ = false;
  int have_plain_ = 0;
  std::vector<Modulus> coeff_;
  Modulus plain_;
};

// ---- context.rs ----
class Context : public detail::Handle<SEALContext_Destroy, detail::no_copy> {
 public:
  Context(const Context&) = delete;
  Context(Context&&) = default;
  // Context::new (context.rs:63-80)
  Context(const EncryptionParameters& params, bool expand_mod_chain, SecurityLevel level) {
    check(SEALContext_Create(params.get_handle(), expand_mod_chain, (int)level, &h_));
  }
  // Context::new_insecure (context.rs:92-103): no security check
  static Context new_insecure(const EncryptionParameters& params, bool expand_mod_chain) { return Context(params, expand_mod_chain, 0); }

 private:
  Context(const EncryptionParameters& params, bool expand_mod_chain, int level) {
    check(SEALContext_Create(params.get_handle(), expand_mod_chain, level, &h_));
  }
};

namespace detail {
// ToBytes / FromBytes (serialization.rs, plaintext_ciphertext.rs:36-120, key_generator.rs:290-760): SEAL 4.0 wire format
template <long (*SaveSize)(void*, uint8_t, int64_t*), long (*Save)(void*, uint8_t*, uint64_t, uint8_t, int64_t*)>
std::vector<uint8_t> save(void* h, uint8_t compr_mode = 2 /* zstd, the crate's CompressionType::ZStd */) {
  int64_t size = 0;
  check(SaveSize(h, compr_mode, &size));
  std::vector<uint8_t> out((size_t)size);
  int64_t written = 0;
  check(Save(h, out.data(), out.size(), compr_mode, &written));
  out.resize((size_t)written);
  return out;
}
}  // namespace detail

// ---- plaintext_ciphertext.rs ----
class Plaintext : public detail::Handle<Plaintext_Destroy, Plaintext_Create5> {
 public:
  using Handle::Handle;
  Plaintext() { check(Plaintext_Create1(nullptr, &h_)); }                                  // Plaintext::new
  static Plaintext from_hex_string(const std::string& hex_poly) {                          // plaintext_ciphertext.rs:180-217
    void* h = nullptr;
    std::string s(hex_poly);
    check(Plaintext_Create4(s.data(), nullptr, &h));
    return Plaintext(h);
  }
  void resize(uint64_t count) { check(Plaintext_Resize(h_, count)); }
  uint64_t len() const {
    uint64_t n = 0;
    check(Plaintext_CoeffCount(h_, &n));
    return n;
  }
  uint64_t get_coefficient(uint64_t i) const {
    uint64_t v = 0;
    check(Plaintext_CoeffAt(h_, i, &v));
    return v;
  }
  void set_coefficient(uint64_t i, uint64_t v) { check(Plaintext_SetCoeffAt(h_, i, v)); }
  bool is_ntt_form() const {
    bool b = false;
    check(Plaintext_IsNTTForm(h_, &b));
    return b;
  }
  std::vector<uint8_t> as_bytes() const { return detail::save<Plaintext_SaveSize, Plaintext_Save>(h_); }
  static Plaintext from_bytes(const Context& ctx, const std::vector<uint8_t>& data) {
    Plaintext p;
    int64_t read = 0;
    check(Plaintext_Load(p.h_, ctx.get_handle(), const_cast<uint8_t*>(data.data()), data.size(), &read));
    return p;
  }
};

class Ciphertext : public detail::Handle<Ciphertext_Destroy, Ciphertext_Create2> {
 public:
  using Handle::Handle;
  Ciphertext() { check(Ciphertext_Create1(nullptr, &h_)); }  // Ciphertext::new (plaintext_ciphertext.rs:337-345)
  uint64_t num_polynomials() const {
    uint64_t v = 0;
    check(Ciphertext_Size(h_, &v));
    return v;
  }
  uint64_t coeff_modulus_size() const {
    uint64_t v = 0;
    check(Ciphertext_CoeffModulusSize(h_, &v));
    return v;
  }
  uint64_t poly_modulus_degree() const {
    uint64_t v = 0;
    check(Ciphertext_PolyModulusDegree(h_, &v));
    return v;
  }
  bool is_ntt_form() const {
    bool b = false;
    check(Ciphertext_IsNTTForm(h_, &b));
    return b;
  }
  // get_coefficient(poly, coeff): the K residues of one coefficient (plaintext_ciphertext.rs:383-410)
  std::vector<uint64_t> get_coefficient(uint64_t poly_index, uint64_t coeff_index) const {
    std::vector<uint64_t> r(coeff_modulus_size());
    check(Ciphertext_GetDataAt2(h_, poly_index, coeff_index, r.data()));
    return r;
  }
  std::vector<uint8_t> as_bytes() const { return detail::save<Ciphertext_SaveSize, Ciphertext_Save>(h_); }
  static Ciphertext from_bytes(const Context& ctx, const std::vector<uint8_t>& data) {
    Ciphertext c;
    int64_t read = 0;
    check(Ciphertext_Load(c.h_, ctx.get_handle(), const_cast<uint8_t*>(data.data()), data.size(), &read));
    return c;
  }
  // hipbfv extension: the raw layout u64[size][K][N] documented at plaintext_ciphertext.rs:303-314
  static Ciphertext from_raw(const Context& ctx, uint64_t size, const std::vector<uint64_t>& words) {
    Ciphertext c;
    check(hipbfv_Ciphertext_Assign(c.h_, ctx.get_handle(), size, words.data()));
    return c;
  }
  std::vector<uint64_t> to_raw() const {
    std::vector<uint64_t> out(num_polynomials() * coeff_modulus_size() * poly_modulus_degree());
    check(hipbfv_Ciphertext_Export(h_, out.data(), out.size()));
    return out;
  }
};

// ---- key_generator.rs ----
class SecretKey : public detail::Handle<SecretKey_Destroy, SecretKey_Create2> {
 public:
  using Handle::Handle;
  SecretKey() { check(SecretKey_Create1(&h_)); }
  std::vector<uint8_t> as_bytes() const { return detail::save<SecretKey_SaveSize, SecretKey_Save>(h_); }
  static SecretKey from_bytes(const Context& ctx, const std::vector<uint8_t>& data) {
    SecretKey k;
    int64_t read = 0;
    check(SecretKey_Load(k.h_, ctx.get_handle(), const_cast<uint8_t*>(data.data()), data.size(), &read));
    return k;
  }
};
class PublicKey : public detail::Handle<PublicKey_Destroy, PublicKey_Create2> {
 public:
  using Handle::Handle;
  PublicKey() { check(PublicKey_Create1(&h_)); }
  std::vector<uint8_t> as_bytes() const { return detail::save<PublicKey_SaveSize, PublicKey_Save>(h_); }
  static PublicKey from_bytes(const Context& ctx, const std::vector<uint8_t>& data) {
    PublicKey k;
    int64_t read = 0;
    check(PublicKey_Load(k.h_, ctx.get_handle(), const_cast<uint8_t*>(data.data()), data.size(), &read));
    return k;
  }
};
// RelinearizationKeys and GaloisKeys are both SEAL KSwitchKeys (key_generator.rs:467-760)
class KSwitchKeys : public detail::Handle<KSwitchKeys_Destroy, KSwitchKeys_Create2> {
 public:
  using Handle::Handle;
  KSwitchKeys() { check(KSwitchKeys_Create1(&h_)); }
  std::vector<uint8_t> as_bytes() const { return detail::save<KSwitchKeys_SaveSize, KSwitchKeys_Save>(h_); }

 protected:
  void load(const Context& ctx, const std::vector<uint8_t>& data) {
    int64_t read = 0;
    check(KSwitchKeys_Load(h_, ctx.get_handle(), const_cast<uint8_t*>(data.data()), data.size(), &read));
  }
};
class RelinearizationKeys : public KSwitchKeys {
 public:
  using KSwitchKeys::KSwitchKeys;
  static RelinearizationKeys from_bytes(const Context& ctx, const std::vector<uint8_t>& data) {
    RelinearizationKeys k;
    k.load(ctx, data);
    return k;
  }
};
class GaloisKeys : public KSwitchKeys {
 public:
  using KSwitchKeys::KSwitchKeys;
  static GaloisKeys from_bytes(const Context& ctx, const std::vector<uint8_t>& data) {
    GaloisKeys k;
    k.load(ctx, data);
    return k;
  }
};

class KeyGenerator : public detail::Handle<KeyGenerator_Destroy, detail::no_copy> {
 public:
  KeyGenerator(const KeyGenerator&) = delete;
  KeyGenerator(KeyGenerator&&) = default;
  explicit KeyGenerator(const Context& ctx) { check(KeyGenerator_Create1(ctx.get_handle(), &h_)); }  // key_generator.rs:32-41
  KeyGenerator(const Context& ctx, const SecretKey& sk) {                                           // new_from_secret_key, :52-63
    check(KeyGenerator_Create2(ctx.get_handle(), sk.get_handle(), &h_));
  }
  SecretKey secret_key() const {
    void* h = nullptr;
    check(KeyGenerator_SecretKey(h_, &h));
    return SecretKey(h);
  }
  PublicKey create_public_key() const {
    void* h = nullptr;
    check(KeyGenerator_CreatePublicKey(h_, false, &h));
    return PublicKey(h);
  }
  RelinearizationKeys create_relinearization_keys() const {
    void* h = nullptr;
    check(KeyGenerator_CreateRelinKeys(h_, false, &h));
    return RelinearizationKeys(h);
  }
  GaloisKeys create_galois_keys() const {  // key_generator.rs:170-184: every power-of-two rotation + the column swap
    void* h = nullptr;
    check(KeyGenerator_CreateGaloisKeysAll(h_, false, &h));
    return GaloisKeys(h);
  }
};

// ---- encoder.rs:40-215 ----
class BFVEncoder : public detail::Handle<BatchEncoder_Destroy, detail::no_copy> {
 public:
  BFVEncoder(const BFVEncoder&) = delete;
  BFVEncoder(BFVEncoder&&) = default;
  explicit BFVEncoder(const Context& ctx) { check(BatchEncoder_Create(ctx.get_handle(), &h_)); }
  size_t get_slot_count() const {
    uint64_t n = 0;
    check(BatchEncoder_GetSlotCount(h_, &n));
    return (size_t)n;
  }
  Plaintext encode_unsigned(const std::vector<uint64_t>& data) const {
    Plaintext p;
    std::vector<uint64_t> d(data);
    check(BatchEncoder_Encode1(h_, d.size(), d.data(), p.get_handle()));
    return p;
  }
  Plaintext encode_signed(const std::vector<int64_t>& data) const {
    Plaintext p;
    std::vector<int64_t> d(data);
    check(BatchEncoder_Encode2(h_, d.size(), d.data(), p.get_handle()));
    return p;
  }
  std::vector<uint64_t> decode_unsigned(const Plaintext& p) const {
    std::vector<uint64_t> out(get_slot_count());
    uint64_t size = 0;
    check(BatchEncoder_Decode1(h_, p.get_handle(), &size, out.data(), nullptr));
    out.resize((size_t)size);
    return out;
  }
  std::vector<int64_t> decode_signed(const Plaintext& p) const {
    std::vector<int64_t> out(get_slot_count());
    uint64_t size = 0;
    check(BatchEncoder_Decode2(h_, p.get_handle(), &size, out.data(), nullptr));
    out.resize((size_t)size);
    return out;
  }
};

// ---- encryptor_decryptor.rs ----
class Encryptor : public detail::Handle<Encryptor_Destroy, detail::no_copy> {
 public:
  Encryptor(const Encryptor&) = delete;
  Encryptor(Encryptor&&) = default;
  static Encryptor with_public_key(const Context& ctx, const PublicKey& pk) {  // :165-183
    Encryptor e;
    check(Encryptor_Create(ctx.get_handle(), pk.get_handle(), nullptr, &e.h_));
    return e;
  }
  static Encryptor with_secret_key(const Context& ctx, const SecretKey& sk) {  // :185-203
    Encryptor e;
    check(Encryptor_Create(ctx.get_handle(), nullptr, sk.get_handle(), &e.h_));
    return e;
  }
  static Encryptor with_public_and_secret_key(const Context& ctx, const PublicKey& pk, const SecretKey& sk) {  // :139-163
    Encryptor e;
    check(Encryptor_Create(ctx.get_handle(), pk.get_handle(), sk.get_handle(), &e.h_));
    return e;
  }
  Ciphertext encrypt(const Plaintext& p) const {  // :238-254
    Ciphertext c;
    check(Encryptor_Encrypt(h_, p.get_handle(), c.get_handle(), nullptr));
    return c;
  }
  Ciphertext encrypt_symmetric(const Plaintext& p) const {  // :416-432
    Ciphertext c;
    check(Encryptor_EncryptSymmetric(h_, p.get_handle(), false, c.get_handle(), nullptr));
    return c;
  }

 private:
  Encryptor() = default;
};

class Decryptor : public detail::Handle<Decryptor_Destroy, detail::no_copy> {
 public:
  Decryptor(const Decryptor&) = delete;
  Decryptor(Decryptor&&) = default;
  Decryptor(const Context& ctx, const SecretKey& sk) { check(Decryptor_Create(ctx.get_handle(), sk.get_handle(), &h_)); }  // :603-616
  Plaintext decrypt(const Ciphertext& c) const {  // :618-632
    Plaintext p;
    check(Decryptor_Decrypt(h_, c.get_handle(), p.get_handle()));
    return p;
  }
  uint32_t invariant_noise_budget(const Ciphertext& c) const {  // :647-658
    int b = 0;
    check(Decryptor_InvariantNoiseBudget(h_, c.get_handle(), &b));
    return (uint32_t)b;
  }
  double invariant_noise(const Ciphertext& c) const {  // the fork's measure, :674-683
    double v = 0;
    check(Decryptor_InvariantNoise(h_, c.get_handle(), &v));
    return v;
  }
};

// ---- trait Evaluator (evaluator.rs:7-280) implemented by BFVEvaluator (bfv_evaluator.rs:27-244, evaluator_base.rs) ----
class BFVEvaluator : public detail::Handle<Evaluator_Destroy, detail::no_copy> {
 public:
  BFVEvaluator(const BFVEvaluator&) = delete;
  BFVEvaluator(BFVEvaluator&&) = default;
  explicit BFVEvaluator(const Context& ctx) { check(Evaluator_Create(ctx.get_handle(), &h_)); }  // bfv_evaluator.rs:27-29

  void negate_inplace(Ciphertext& a) const { check(Evaluator_Negate(h_, a.get_handle(), a.get_handle())); }
  Ciphertext negate(const Ciphertext& a) const {
    Ciphertext out;
    check(Evaluator_Negate(h_, a.get_handle(), out.get_handle()));
    return out;
  }
  void add_inplace(Ciphertext& a, const Ciphertext& b) const { check(Evaluator_Add(h_, a.get_handle(), b.get_handle(), a.get_handle())); }
  Ciphertext add(const Ciphertext& a, const Ciphertext& b) const {
    Ciphertext out;
    check(Evaluator_Add(h_, a.get_handle(), b.get_handle(), out.get_handle()));
    return out;
  }
  Ciphertext add_many(const std::vector<Ciphertext>& a) const {  // evaluator_base.rs:117-137
    std::vector<void*> hs;
    for (const Ciphertext& c : a) hs.push_back(c.get_handle());
    Ciphertext out;
    check(Evaluator_AddMany(h_, hs.size(), hs.data(), out.get_handle()));
    return out;
  }
  Ciphertext multiply_many(const std::vector<Ciphertext>& a, const RelinearizationKeys& rk) const {  // evaluator_base.rs:139-166
    std::vector<void*> hs;
    for (const Ciphertext& c : a) hs.push_back(c.get_handle());
    Ciphertext out;
    check(Evaluator_MultiplyMany(h_, hs.size(), hs.data(), rk.get_handle(), out.get_handle(), nullptr));
    return out;
  }
  void sub_inplace(Ciphertext& a, const Ciphertext& b) const { check(Evaluator_Sub(h_, a.get_handle(), b.get_handle(), a.get_handle())); }
  Ciphertext sub(const Ciphertext& a, const Ciphertext& b) const {
    Ciphertext out;
    check(Evaluator_Sub(h_, a.get_handle(), b.get_handle(), out.get_handle()));
    return out;
  }
  void multiply_inplace(Ciphertext& a, const Ciphertext& b) const {  // evaluator_base.rs:184-196
    check(Evaluator_Multiply(h_, a.get_handle(), b.get_handle(), a.get_handle(), nullptr));
  }
  Ciphertext multiply(const Ciphertext& a, const Ciphertext& b) const {  // evaluator_base.rs:198-212
    Ciphertext out;
    check(Evaluator_Multiply(h_, a.get_handle(), b.get_handle(), out.get_handle(), nullptr));
    return out;
  }
  void square_inplace(Ciphertext& a) const { check(Evaluator_Square(h_, a.get_handle(), a.get_handle(), nullptr)); }
  Ciphertext square(const Ciphertext& a) const {
    Ciphertext out;
    check(Evaluator_Square(h_, a.get_handle(), out.get_handle(), nullptr));
    return out;
  }
  Ciphertext mod_switch_to_next(const Ciphertext& a) const {  // evaluator_base.rs:238-252
    Ciphertext out;
    check(Evaluator_ModSwitchToNext1(h_, a.get_handle(), out.get_handle(), nullptr));
    return out;
  }
  void mod_switch_to_next_inplace(Ciphertext& a) const { check(Evaluator_ModSwitchToNext1(h_, a.get_handle(), a.get_handle(), nullptr)); }
  Plaintext mod_switch_to_next_plaintext(const Plaintext& a) const {
    Plaintext out;
    check(Evaluator_ModSwitchToNext2(h_, a.get_handle(), out.get_handle()));
    return out;
  }
  void mod_switch_to_next_inplace_plaintext(Plaintext& a) const { check(Evaluator_ModSwitchToNext2(h_, a.get_handle(), a.get_handle())); }
  Ciphertext exponentiate(const Ciphertext& a, uint64_t exponent, const RelinearizationKeys& rk) const {  // evaluator_base.rs:290-312
    Ciphertext out;
    check(Evaluator_Exponentiate(h_, a.get_handle(), exponent, rk.get_handle(), out.get_handle(), nullptr));
    return out;
  }
  void exponentiate_inplace(Ciphertext& a, uint64_t exponent, const RelinearizationKeys& rk) const {
    check(Evaluator_Exponentiate(h_, a.get_handle(), exponent, rk.get_handle(), a.get_handle(), nullptr));
  }
  Ciphertext add_plain(const Ciphertext& a, const Plaintext& b) const {
    Ciphertext out;
    check(Evaluator_AddPlain(h_, a.get_handle(), b.get_handle(), out.get_handle()));
    return out;
  }
  void add_plain_inplace(Ciphertext& a, const Plaintext& b) const { check(Evaluator_AddPlain(h_, a.get_handle(), b.get_handle(), a.get_handle())); }
  Ciphertext sub_plain(const Ciphertext& a, const Plaintext& b) const {
    Ciphertext out;
    check(Evaluator_SubPlain(h_, a.get_handle(), b.get_handle(), out.get_handle()));
    return out;
  }
  void sub_plain_inplace(Ciphertext& a, const Plaintext& b) const { check(Evaluator_SubPlain(h_, a.get_handle(), b.get_handle(), a.get_handle())); }
  Ciphertext multiply_plain(const Ciphertext& a, const Plaintext& b) const {
    Ciphertext out;
    check(Evaluator_MultiplyPlain(h_, a.get_handle(), b.get_handle(), out.get_handle(), nullptr));
    return out;
  }
  void multiply_plain_inplace(Ciphertext& a, const Plaintext& b) const {
    check(Evaluator_MultiplyPlain(h_, a.get_handle(), b.get_handle(), a.get_handle(), nullptr));
  }
  void relinearize_inplace(Ciphertext& a, const RelinearizationKeys& rk) const {  // bfv_evaluator.rs:148-164
    check(Evaluator_Relinearize(h_, a.get_handle(), rk.get_handle(), a.get_handle(), nullptr));
  }
  Ciphertext relinearize(const Ciphertext& a, const RelinearizationKeys& rk) const {  // bfv_evaluator.rs:166-184
    Ciphertext out;
    check(Evaluator_Relinearize(h_, a.get_handle(), rk.get_handle(), out.get_handle(), nullptr));
    return out;
  }
  // steps > 0 rotates the rows left, < 0 right (bfv_evaluator.rs:186-226)
  Ciphertext rotate_rows(const Ciphertext& a, int steps, const GaloisKeys& gk) const {
    Ciphertext out;
    check(Evaluator_RotateRows(h_, a.get_handle(), steps, gk.get_handle(), out.get_handle(), nullptr));
    return out;
  }
  void rotate_rows_inplace(Ciphertext& a, int steps, const GaloisKeys& gk) const {
    check(Evaluator_RotateRows(h_, a.get_handle(), steps, gk.get_handle(), a.get_handle(), nullptr));
  }
  Ciphertext rotate_columns(const Ciphertext& a, const GaloisKeys& gk) const {  // bfv_evaluator.rs:228-244
    Ciphertext out;
    check(Evaluator_RotateColumns(h_, a.get_handle(), gk.get_handle(), out.get_handle(), nullptr));
    return out;
  }
  void rotate_columns_inplace(Ciphertext& a, const GaloisKeys& gk) const {
    check(Evaluator_RotateColumns(h_, a.get_handle(), gk.get_handle(), a.get_handle(), nullptr));
  }
};

}  // namespace seal_fhe
