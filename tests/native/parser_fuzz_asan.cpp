// tests/native/parser_fuzz_asan.cpp -- the host-side parsers (SEAL wire decoders, program JSON loader) built with
// AddressSanitizer + UndefinedBehaviorSanitizer and fed mutated inputs through their internal C++ entry points: reads past a
// buffer, overflows and leaks that a plain run survives are reported.  Built and run by tests/test_decoder_fuzz_cpu.py:
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Isunscreen_amd/csrc -x c++ \
//       tests/native/parser_fuzz_asan.cpp sunscreen_amd/csrc/wire.cpp sunscreen_amd/csrc/program.cpp sunscreen_amd/csrc/program_plan.cpp -ldl -Wl,--unresolved-symbols=ignore-all
// (program.cpp / program_plan.cpp also hold the executors, whose device calls stay unresolved and uncalled).  argv: seed, iterations, JSON seed files.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "wire.hpp"
#include "program.hpp"
using namespace hipbfv;
static std::mt19937_64 rng;
static std::vector<uint8_t> mutate(std::vector<uint8_t> b) {
  int rounds = 1 + rng() % 5;
  for (int r = 0; r < rounds && !b.empty(); r++) {
    switch (rng() % 6) {
      case 0: { size_t i = (rng() % 10 < 7) ? rng() % std::min<size_t>(b.size(), 96) : rng() % b.size(); b[i] ^= 1u << (rng() % 8); break; }
      case 1: { size_t i = rng() % std::max<size_t>(1, b.size() > 8 ? b.size() - 8 : 1); unsigned long long vals[] = {0, 1, 0xFF, 1ull << 31, (1ull << 63) - 1, ~0ull, b.size(), b.size() * 8};
                unsigned long long v = vals[rng() % 8]; if (i + 8 <= b.size()) memcpy(&b[i], &v, 8); break; }
      case 2: b.resize(rng() % b.size()); break;
      case 3: { size_t i = rng() % b.size(), j = rng() % b.size(); if (i > j) std::swap(i, j); std::vector<uint8_t> s(b.begin() + i, b.begin() + std::min(j, i + 4096)); b.insert(b.begin() + i, s.begin(), s.end()); break; }
      case 4: { size_t i = rng() % b.size(); for (size_t k = i; k < std::min(b.size(), i + 16); k++) b[k] = (uint8_t)rng(); break; }
      default: { size_t i = rng() % b.size(), j = rng() % b.size(); if (i > j) std::swap(i, j); b.erase(b.begin() + i, b.begin() + j); }
    }
  }
  return b;
}
int main(int argc, char** argv) {
  rng.seed(argc > 1 ? atoi(argv[1]) : 1);
  const int iters = argc > 2 ? atoi(argv[2]) : 20000;
  std::vector<std::vector<uint8_t>> cts, pts, kss;
  for (int compr : {0, 2}) {
    if (compr == 2 && !wire_zstd_available()) continue;
    std::vector<unsigned long long> data(2 * 2 * 1024);
    for (auto& x : data) x = rng() >> 24;
    uint8_t pid[32] = {1, 2, 3};
    std::vector<uint8_t> out;
    if (wire_pack_ciphertext(pid, true, 2, 1024, 2, data.data(), compr, &out) == 0) cts.push_back(out);
    std::vector<unsigned long long> coeffs(1024, 5);
    if (wire_pack_plaintext(pid, coeffs.data(), coeffs.size(), compr, &out) == 0) pts.push_back(out);
    std::vector<std::vector<const unsigned long long*>> keys(3);
    keys[0] = {data.data(), data.data()};
    keys[2] = {data.data()};
    if (wire_pack_kswitch(pid, 1024, 2, keys, compr, &out) == 0) kss.push_back(out);
  }
  long acc = 0;
  for (int it = 0; it < iters; it++) {
    size_t used = 0;
    { WireCiphertext c; auto m = mutate(cts[it % cts.size()]); acc += wire_unpack_ciphertext(m.data(), m.size(), &c, &used, (it & 1) ? 1 << 20 : 0) == 0; }
    { WirePlaintext p; auto m = mutate(pts[it % pts.size()]); acc += wire_unpack_plaintext(m.data(), m.size(), &p, &used, (it & 1) ? 1 << 20 : 0) == 0; }
    { WireKSwitchKeys k; auto m = mutate(kss[it % kss.size()]); acc += wire_unpack_kswitch(m.data(), m.size(), &k, &used, (it & 1) ? 1 << 22 : 0, (it & 2) ? 64 : 0) == 0; }
  }
  // JSON programs: seeds read from files given on the command line
  for (int a = 3; a < argc; a++) {
    FILE* f = fopen(argv[a], "rb"); if (!f) continue;
    std::string s; char buf[4096]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n); fclose(f);
    for (int it = 0; it < iters / 4; it++) {
      std::vector<uint8_t> m = mutate(std::vector<uint8_t>(s.begin(), s.end()));
      Program p; std::string err;
      // exact-size heap copy without a terminator: ASan sees any read past the end
      char* exact = (char*)malloc(m.size() ? m.size() : 1); memcpy(exact, m.data(), m.size());
      acc += p.load_json(exact, m.size(), &err) == 0;
      free(exact);
    }
  }
  printf("asan fuzz ok %ld\n", acc);
  return 0;
}
