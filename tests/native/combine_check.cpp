// tests/native/combine_check.cpp -- many host threads on ONE evaluator handle, mixed operation kinds (multiply, relinearize, rotate, square, add, sub, add_plain, sub_plain, multiply_plain incl. a monomial), the way
// sunscreen_runtime dispatches graph nodes from a rayon pool (run.rs:415-469).  Concurrent calls are combined into batched
// launches inside the library (capi.cpp Combiner): whatever got combined with whatever, every thread must get exactly the
// bits it gets alone, and a thread whose result is transparent must be the only one that sees an error.
//
//   g++ -O1 -std=c++17 -Iinclude tests/native/combine_check.cpp -Lsunscreen_amd/lib -lhipbfv -Wl,-rpath,<lib dir> -lpthread
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

#include "hipbfv.hpp"

using namespace seal_fhe;

static std::vector<uint64_t> words(const Ciphertext& c) { return c.to_raw(); }

int main() {
  try {
    const uint64_t n = 4096;
    auto params = BfvEncryptionParametersBuilder()
                      .set_poly_modulus_degree(n)
                      .set_coefficient_modulus(CoeffModulus::bfv_default(n))
                      .set_plain_modulus(PlainModulus::batching(n, 16))
                      .build();
    Context ctx(params, true, SecurityLevel::TC128);
    KeyGenerator gen(ctx);
    const PublicKey pk = gen.create_public_key();
    const RelinearizationKeys rk = gen.create_relinearization_keys();
    const GaloisKeys gk = gen.create_galois_keys();
    BFVEncoder encoder(ctx);
    const Encryptor encryptor = Encryptor::with_public_key(ctx, pk);
    BFVEvaluator ev(ctx);
    const int T = 24, iters = 12;
    std::vector<Ciphertext> a, b;
    std::vector<Plaintext> pt;
    const Plaintext mono = Plaintext::from_hex_string("3x^5");
    std::vector<int64_t> x(encoder.get_slot_count());
    for (int i = 0; i < T; i++) {
      for (size_t j = 0; j < x.size(); j++) x[j] = (int64_t)((j * (i + 3)) % 23) - 11;
      a.push_back(encryptor.encrypt(encoder.encode_signed(x)));
      for (size_t j = 0; j < x.size(); j++) x[j] = (int64_t)((j + i) % 7) - 3;
      b.push_back(encryptor.encrypt(encoder.encode_signed(x)));
      for (size_t j = 0; j < x.size(); j++) x[j] = (int64_t)((j * 5 + i) % 9) - 4;
      pt.push_back(encoder.encode_signed(x));
    }
    // what every thread gets when it is alone
    std::vector<std::vector<uint64_t>> want_rel(T), want_rot(T), want_sq(T), want_add(T), want_sub(T), want_ap(T), want_sp(T), want_mp(T), want_mono(T);

    for (int i = 0; i < T; i++) {
      Ciphertext m = ev.multiply(a[i], b[i]);
      check(Evaluator_Relinearize(ev.get_handle(), m.get_handle(), rk.get_handle(), m.get_handle(), nullptr));
      want_rel[i] = words(m);
      want_rot[i] = words(ev.rotate_rows(a[i], 1 + (i % 3), gk));
      want_sq[i] = words(ev.square(b[i]));
      want_add[i] = words(ev.add(a[i], b[i]));
      want_sub[i] = words(ev.sub(a[i], b[i]));
      want_ap[i] = words(ev.add_plain(a[i], pt[i]));
      want_sp[i] = words(ev.sub_plain(a[i], pt[i]));
      want_mp[i] = words(ev.multiply_plain(a[i], pt[i]));
      want_mono[i] = words(ev.multiply_plain(b[i], mono));
    }
    // an operand whose second polynomial is zero (imported raw: the evaluator itself would refuse to produce it): its product
    // with another such operand is (a0 b0, 0, 0) -- transparent
    std::vector<uint64_t> raw = a[0].to_raw();
    for (size_t j = raw.size() / 2; j < raw.size(); j++) raw[j] = 0;
    Ciphertext zero = Ciphertext::from_raw(ctx, 2, raw);
    std::atomic<int> bad{0}, transparent_seen{0};
    std::vector<std::thread> ths;
    for (int i = 0; i < T; i++)
      ths.emplace_back([&, i] {
        int last_kind = -1;
        try {
          for (int it = 0; it < iters; it++) {
            const int what = (i + it) % 6;
            last_kind = what;
            if (what == 0 || what == 1) {
              Ciphertext m = ev.multiply(a[i], b[i]);
              check(Evaluator_Relinearize(ev.get_handle(), m.get_handle(), rk.get_handle(), m.get_handle(), nullptr));
              if (words(m) != want_rel[i]) bad++, std::fprintf(stderr, "thread %d it %d: multiply+relinearize differs\n", i, it);
            } else if (what == 2) {
              if (words(ev.rotate_rows(a[i], 1 + (i % 3), gk)) != want_rot[i]) bad++, std::fprintf(stderr, "thread %d it %d: rotation differs\n", i, it);
            } else if (what == 5) {
              if (words(ev.add_plain(a[i], pt[i])) != want_ap[i] || words(ev.sub_plain(a[i], pt[i])) != want_sp[i] ||
                  words(ev.multiply_plain(a[i], pt[i])) != want_mp[i] || words(ev.multiply_plain(b[i], mono)) != want_mono[i])
                bad++, std::fprintf(stderr, "thread %d it %d: a plaintext operation differs\n", i, it);
            } else if (what == 4) {
              if (words(ev.add(a[i], b[i])) != want_add[i] || words(ev.sub(a[i], b[i])) != want_sub[i]) bad++, std::fprintf(stderr, "thread %d it %d: add / sub differs\n", i, it);
            } else {
              if (words(ev.square(b[i])) != want_sq[i]) bad++, std::fprintf(stderr, "thread %d it %d: square differs\n", i, it);
            }
            if (i == 5) {  // this thread also asks for a transparent product: it alone must be refused
              Ciphertext out;
              const long hr = Evaluator_Multiply(ev.get_handle(), zero.get_handle(), zero.get_handle(), out.get_handle(), nullptr);
              if (hr == HIPBFV_COR_E_INVALIDOPERATION) transparent_seen++;
              else bad++, std::fprintf(stderr, "thread %d it %d: transparent product returned %lx\n", i, it, (unsigned long)hr);
            }
          }
        } catch (const Error& e) {
          std::fprintf(stderr, "thread %d (last op kind %d): %s\n", i, last_kind, e.what());
          bad++;
        }
      });
    for (auto& t : ths) t.join();
    if (bad.load() || transparent_seen.load() != iters) {
      std::printf("FAILED bad=%d transparent=%d\n", bad.load(), transparent_seen.load());
      return 2;
    }
    std::printf("combine ok\n");
    return 0;
  } catch (const Error& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
