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
    for (size_t i = 0; i < x.size(); i++) x[i] = (int64_t)(i % 31) - 15, y[i] = 5;
    const int max_threads = 64;
    std::vector<Ciphertext> a, b, out;
    for (int i = 0; i < max_threads; i++) {
      a.push_back(encryptor.encrypt(encoder.encode_signed(x)));
      b.push_back(encryptor.encrypt(encoder.encode_signed(y)));
      out.emplace_back();
    }
    std::printf("{\"n\": %llu, \"multiply_relinearize_ops_per_s_by_threads\": {", (unsigned long long)n);
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
    const std::vector<int64_t> got = encoder.decode_signed(decryptor.decrypt(out[0]));
    for (size_t i = 0; i < got.size(); i++)
      if (got[i] != x[i] * y[i]) return 3;
  } catch (const Error& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
