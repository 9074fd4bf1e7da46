// tests/native/rng_check.cpp -- the block function of sunscreen_amd/csrc/rng.hpp against RFC 8439's known answer, and the
// properties the seed derivation promises.  Host build (g++) of the same header the kernels include.
#include <cstdio>
#include <initializer_list>
#include <cstring>

#include "../../sunscreen_amd/csrc/rng.hpp"

using namespace hipbfv;

int main() {
  // RFC 8439 section 2.3.2: key 00..1f, block counter 1, nonce 00:00:00:09 00:00:00:4a 00:00:00:00
  RngKey key;
  for (int i = 0; i < 8; i++) key.k[i] = (uint32_t)(4 * i) | (uint32_t)(4 * i + 1) << 8 | (uint32_t)(4 * i + 2) << 16 | (uint32_t)(4 * i + 3) << 24;
  uint32_t out[16];
  chacha20_block<16>(key, 1u, 0x09000000u, 0x4a000000u, 0u, out);
  const uint32_t expect[16] = {0xe4e7f110u, 0x15593bd1u, 0x1fdd0f50u, 0xc47120a3u, 0xc7f4d1c7u, 0x0368c033u, 0x9aaa2204u, 0x4e6cd4c3u,
                               0x466482d2u, 0x09aa9f07u, 0x05d7c214u, 0xa2028bd9u, 0xd19c12b5u, 0xb94e16deu, 0xe883d0cbu, 0x4e3c50a2u};
  if (std::memcmp(out, expect, sizeof(out)) != 0) {
    std::printf("chacha20 block differs from RFC 8439 2.3.2\n");
    return 1;
  }
  uint32_t first5[5];
  chacha20_block<5>(key, 1u, 0x09000000u, 0x4a000000u, 0u, first5);
  if (std::memcmp(first5, expect, sizeof(first5)) != 0) return 2;  // a truncated read is a prefix of the block
  // seed derivation: both halves of the 512-bit seed reach both keys; secret and pub differ; deterministic
  uint8_t seed[64];
  for (int i = 0; i < 64; i++) seed[i] = (uint8_t)(i * 7 + 3);
  const RngSeed a = rng_seed_from_512(seed);
  const RngSeed again = rng_seed_from_512(seed);
  if (std::memcmp(&a, &again, sizeof(a)) != 0) return 3;
  if (std::memcmp(&a.secret, &a.pub, sizeof(RngKey)) == 0) return 4;
  for (int flip : {0, 31, 32, 63}) {
    uint8_t s2[64];
    std::memcpy(s2, seed, 64);
    s2[flip] ^= 1;
    const RngSeed b = rng_seed_from_512(s2);
    if (std::memcmp(&a.secret, &b.secret, sizeof(RngKey)) == 0 || std::memcmp(&a.pub, &b.pub, sizeof(RngKey)) == 0) return 5;
  }
  const RngSeed t1 = rng_seed_from_u64_for_tests(1), t2 = rng_seed_from_u64_for_tests(2);
  if (std::memcmp(&t1, &t2, sizeof(t1)) == 0) return 6;
  // dump what the GPU tests recompute: the keys of test seed 0x1234 and one block under its secret key
  const RngSeed t = rng_seed_from_u64_for_tests(0x1234);
  uint32_t blk[16];
  chacha20_block<16>(t.secret, 7u, 0u, 0u, 0x5Eu, blk);
  std::printf("rng ok %08x %08x %08x\n", t.secret.k[0], t.pub.k[0], blk[0]);
  return 0;
}
