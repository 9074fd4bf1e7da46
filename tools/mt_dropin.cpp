// tools/mt_dropin.cpp -- throughput of the handle-level (one ciphertext per call) C ABI when several host threads share
// one evaluator, the way sunscreen_runtime dispatches ready graph nodes from a rayon pool (run.rs:415-469): every thread runs
// multiply + relinearize on its own ciphertexts; prints ops/s for 1, 2, 4 ... threads.  No Python, no GIL: what is measured
// is the library (locks, stream handling, buffer cache) and the device.
//
//   g++ -O2 -std=c++17 -Iinclude tools/mt_dropin.cpp -Lsunscreen_amd/lib -lhipbfv -Wl,-rpath,$PWD/sunscreen_amd/lib -lpthread -o /tmp/mt_dropin
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "hipbfv.hpp"

using namespace seal_fhe;

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 8192;
  const double seconds = argc > 2 ? std::atof(argv[2]) : 1.5;
  try {
    auto params = BfvEncryptionParametersBuilder()
                      .set_poly_modulus_degree(n)
                      .set_coefficient_modulus(CoeffModulus::bfv_default(n))
                      .set_plain_modulus(PlainModulus::batching(n, 20))
                      .build();
    Context ctx(params, true, SecurityLevel::TC128);
    KeyGenerator gen(ctx);
    const PublicKey pk = gen.create_public_key();
    const RelinearizationKeys rk = gen.create_relinearization_keys();
    BFVEncoder encoder(ctx);
    const Encryptor encryptor = Encryptor::with_public_key(ctx, pk);
    Decryptor decryptor(ctx, gen.secret_key());
    BFVEvaluator ev(ctx);
    std::vector<int64_t> x(encoder.get_slot_count()), y(x.size());
    for (size_t i = 0; i < x.size(); i++) x[i] = (int64_t)(i % 7) - 3, y[i] = 2;  // small values: chi_sq raises them to the fourth power
    const int max_threads = 64;
    std::vector<Ciphertext> a, b, out;
    for (int i = 0; i < max_threads; i++) {
      a.push_back(encryptor.encrypt(encoder.encode_signed(x)));
      b.push_back(encryptor.encrypt(encoder.encode_signed(y)));
      out.emplace_back();
    }
    const bool chi = argc > 3 && std::string(argv[3]) == "chi_sq";
    // chi_sq: every thread evaluates examples/chi_sq (main.rs:59-88: 6 multiply + relinearize of which 4 squares, 8 add, 1 sub)
    // node by node through the handle-level calls, the way run.rs walks the graph
    auto mulrel = [&](const Ciphertext& p, const Ciphertext& q) {
      Ciphertext m = ev.multiply(p, q);
      check(Evaluator_Relinearize(ev.get_handle(), m.get_handle(), rk.get_handle(), m.get_handle(), nullptr));
      return m;
    };
    auto chi_sq = [&](const Ciphertext& n0, const Ciphertext& n1, const Ciphertext& n2) {
      const Ciphertext xx = ev.add(ev.add(n0, n0), n1), yy = ev.add(ev.add(n2, n2), n1);
      Ciphertext n02 = mulrel(n0, n2);
      n02 = ev.add(n02, n02);
      n02 = ev.add(n02, n02);
      const Ciphertext al = ev.sub(n02, mulrel(n1, n1));
      const Ciphertext alpha = mulrel(al, al);
      Ciphertext b1 = mulrel(xx, xx);
      b1 = ev.add(b1, b1);
      const Ciphertext b2 = mulrel(xx, yy);
      Ciphertext b3 = mulrel(yy, yy);
      b3 = ev.add(b3, b3);
      return alpha;
    };
    (void)chi_sq(a[0], b[0], a[1]);  // warm-up: the first launch of every kernel loads its code object
    std::printf("{\"n\": %llu, \"%s_per_s_by_threads\": {", (unsigned long long)n, chi ? "chi_sq_programs" : "multiply_relinearize_ops");
    bool first = true;
    for (int nt : {1, 2, 4, 8, 16, 32, 64}) {
      std::atomic<long> total{0};
      std::atomic<bool> stop{false};
      std::vector<std::thread> ths;
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < nt; i++)
        ths.emplace_back([&, i] {
          long cnt = 0;
          while (!stop.load(std::memory_order_relaxed)) {
            if (chi) {
              chi_sq(a[i], b[i], a[(i + 1) % max_threads]);
              cnt++;
              continue;
            }
            check(Evaluator_Multiply(ev.get_handle(), a[i].get_handle(), b[i].get_handle(), out[i].get_handle(), nullptr));
            check(Evaluator_Relinearize(ev.get_handle(), out[i].get_handle(), rk.get_handle(), out[i].get_handle(), nullptr));
            cnt++;
          }
          total += cnt;
        });
      std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
      stop = true;
      for (auto& t : ths) t.join();
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      std::printf("%s\"%d\": %.0f", first ? "" : ", ", nt, total.load() / dt);
      first = false;
      std::fflush(stdout);
    }
    std::printf("}}\n");
    // the last result of thread 0 still decrypts to the product
    if (!chi) {
      const std::vector<int64_t> got = encoder.decode_signed(decryptor.decrypt(out[0]));
      for (size_t i = 0; i < got.size(); i++)
        if (got[i] != x[i] * y[i]) return 3;
    }
  } catch (const Error& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
