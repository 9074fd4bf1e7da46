// examples/simple_multiply.cpp -- the reference's examples/simple_multiply (one BFV ct x ct multiply + relinearize,
// examples/simple_multiply/src/main.rs:57-80: 15 * 5 = 75) written against include/hipbfv.hpp the way the seal_fhe crate's
// own tests use its API (seal_fhe/src/bfv_evaluator.rs:322-420): keygen, batch-encode, encrypt, multiply, relinearize,
// rotate, decrypt, all through libhipbfv.so on the GPU.
//
//   g++ -std=c++17 -Iinclude examples/simple_multiply.cpp -Lsunscreen_amd/lib -lhipbfv -Wl,-rpath,$PWD/sunscreen_amd/lib -o simple_multiply
//   ./simple_multiply             # needs an MI355X
//   ./simple_multiply --host-only # the parts of the surface that never touch the device (parameters, errors)
#include <cstdio>
#include <cstring>

#include "hipbfv.hpp"

using namespace seal_fhe;

static int host_only() {
  // modulus.rs:279-313 known answers: CoeffModulus::create(8192, [50, 30, 30, 50, 50]) and PlainModulus::batching
  const auto m = CoeffModulus::create(8192, {50, 30, 30, 50, 50});
  if (m.size() != 5 || m[0].value() != 1125899905744897ull || m[1].value() != 1073643521ull) return 10;
  if (PlainModulus::batching(8192, 20).value() != 1032193ull) return 11;
  const auto def = CoeffModulus::bfv_default(8192);
  if (def.size() != 5 || def[0].value() != 0x7fffffd8001ull) return 12;
  if (CoeffModulus::max_bit_count(8192) != 218) return 13;
  auto params = BfvEncryptionParametersBuilder().set_poly_modulus_degree(8192).set_coefficient_modulus(CoeffModulus::bfv_default(8192)).set_plain_modulus_u64(1234).build();
  if (params.get_poly_modulus_degree() != 8192 || params.get_plain_modulus().value() != 1234 || params.get_coefficient_modulus().size() != 5) return 14;
  if (params.get_scheme() != SchemeType::Bfv) return 15;
  // builder errors (encryption_parameters.rs:270-300) and the HRESULT mapping (error.rs:65-78)
  try {
    BfvEncryptionParametersBuilder().set_poly_modulus_degree(1024).build();
    return 16;
  } catch (const Error& e) {
    if (e.kind() != Error::InvalidArgument) return 17;
  }
  try {
    check(Evaluator_Negate(nullptr, nullptr, nullptr));
    return 18;
  } catch (const Error& e) {
    if (e.kind() != Error::InvalidPointer) return 19;
  }
  // deep copies and moves keep one owner per handle
  Modulus a(97), b(a), c(std::move(a));
  if (b.value() != 97 || c.value() != 97 || a.get_handle() != nullptr) return 20;
  Plaintext p = Plaintext::from_hex_string("7FFx^3 + 1x^1 + 3");
  if (p.len() != 4 || p.get_coefficient(3) != 0x7FF || p.get_coefficient(0) != 3) return 21;
  Plaintext q(p);
  q.set_coefficient(0, 5);
  if (p.get_coefficient(0) != 3 || q.get_coefficient(0) != 5) return 22;
  std::printf("host-only ok\n");
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && !std::strcmp(argv[1], "--host-only")) return host_only();
  try {
    const uint64_t n = 8192;
    auto params = BfvEncryptionParametersBuilder()
                      .set_poly_modulus_degree(n)
                      .set_coefficient_modulus(CoeffModulus::bfv_default(n))
                      .set_plain_modulus(PlainModulus::batching(n, 20))
                      .build();
    Context ctx(params, true, SecurityLevel::TC128);
    KeyGenerator gen(ctx);
    const SecretKey sk = gen.secret_key();
    const PublicKey pk = gen.create_public_key();
    const RelinearizationKeys rk = gen.create_relinearization_keys();
    const GaloisKeys gk = gen.create_galois_keys();
    BFVEncoder encoder(ctx);
    const Encryptor encryptor = Encryptor::with_public_key(ctx, pk);
    Decryptor decryptor(ctx, sk);
    BFVEvaluator evaluator(ctx);

    std::vector<int64_t> x(encoder.get_slot_count()), y(x.size());
    for (size_t i = 0; i < x.size(); i++) x[i] = (int64_t)(i % 31) - 15, y[i] = 5;
    x[0] = 15;
    const Ciphertext cx = encryptor.encrypt(encoder.encode_signed(x));
    const Ciphertext cy = encryptor.encrypt(encoder.encode_signed(y));
    const uint32_t fresh = decryptor.invariant_noise_budget(cx);

    Ciphertext prod = evaluator.multiply(cx, cy);
    if (prod.num_polynomials() != 3) return 2;
    evaluator.relinearize_inplace(prod, rk);
    if (prod.num_polynomials() != 2) return 3;
    const std::vector<int64_t> z = encoder.decode_signed(decryptor.decrypt(prod));
    for (size_t i = 0; i < x.size(); i++)
      if (z[i] != x[i] * 5) return 4;

    // x << 1 on the rows, then the column swap, then back through the wire format
    const Ciphertext rot = evaluator.rotate_rows(cx, 1, gk);
    const std::vector<int64_t> r = encoder.decode_signed(decryptor.decrypt(rot));
    const size_t half = x.size() / 2;
    for (size_t i = 0; i < half; i++)
      if (r[i] != x[(i + 1) % half] || r[half + i] != x[half + (i + 1) % half]) return 5;
    const Ciphertext back = Ciphertext::from_bytes(ctx, evaluator.rotate_columns(cx, gk).as_bytes());
    const std::vector<int64_t> s = encoder.decode_signed(decryptor.decrypt(back));
    for (size_t i = 0; i < half; i++)
      if (s[i] != x[half + i] || s[half + i] != x[i]) return 6;

    // transparent result is an error, as with the crate's default features (sunscreen/tests/features.rs:8-34)
    Plaintext zero;
    zero.resize(1);
    try {
      (void)evaluator.multiply_plain(cx, zero);
      return 7;
    } catch (const Error& e) {
      if (e.kind() != Error::InternalError) return 8;
    }
    std::printf("simple_multiply: 15 * 5 = %lld; noise budget %u -> %u bits; rotations and the wire format round-trip\n", (long long)z[0], fresh,
                decryptor.invariant_noise_budget(prod));
    return 0;
  } catch (const Error& e) {
    std::fprintf(stderr, "hipbfv error: %s (0x%lx)\n", e.what(), (unsigned long)e.hresult());
    return 1;
  }
}
