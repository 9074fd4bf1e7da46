// sunscreen_amd/csrc/rng.hpp -- the generator behind every sampled polynomial (secret keys, encryption randomness, error
// polynomials, the uniform halves of public / relinearisation / Galois keys): ChaCha20 used as a counter-based PRF.
//
// SEAL draws from a 512-bit-seeded Blake2xb / SHAKE256 stream (UniformRandomGeneratorFactory); a GPU wants random access
// instead of a stream -- thread (coefficient x, operation op) must produce ITS values without anyone else's -- so the
// block function is evaluated at the position: 16 output words = ChaCha20_block(key, input = (x, op_lo, op_hi, domain)).
// Bit-for-bit reproduction of SEAL's stream is not a goal (no reference test pins it); the security
// level is: 256-bit keys, the 20-round block function (RFC 8439's, with the 128-bit input where RFC 8439 puts counter ||
// nonce), independent keys for secret and for published material.
//
// A seed is 512 bits (what SEAL's prng_seed_type and the fork's *SetSeed entry points carry).  It is split into two
// 256-bit ChaCha keys by a domain-separated derivation that uses all 512 bits for each:
//   secret -- ternary secrets, encryption randomness u, every error polynomial: never leaves the device
//   pub    -- the uniform polynomial `a` of public keys, key-switching keys and symmetric ciphertexts: published
// so that what an adversary sees (`a`) is produced under a key that generates nothing secret.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define HIPBFV_RNG_HD __host__ __device__ __forceinline__
#else
#define HIPBFV_RNG_HD inline
#endif

namespace hipbfv {

struct RngKey {
  uint32_t k[8];
};

struct RngSeed {
  RngKey secret;
  RngKey pub;
};

HIPBFV_RNG_HD uint32_t rng_rotl(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }

#define HIPBFV_QR(a, b, c, d) \
  a += b; d ^= a; d = rng_rotl(d, 16); \
  c += d; b ^= c; b = rng_rotl(b, 12); \
  a += b; d ^= a; d = rng_rotl(d, 8);  \
  c += d; b ^= c; b = rng_rotl(b, 7);

// out[0..NOUT) = the first NOUT words of the ChaCha20 block (20 rounds + feed-forward) for `key` at input (i0,i1,i2,i3)
template <int NOUT>
HIPBFV_RNG_HD void chacha20_block(const RngKey& key, uint32_t i0, uint32_t i1, uint32_t i2, uint32_t i3, uint32_t (&out)[NOUT]) {
  static_assert(NOUT >= 1 && NOUT <= 16, "a block has 16 words");
  const uint32_t s0 = 0x61707865u, s1 = 0x3320646eu, s2 = 0x79622d32u, s3 = 0x6b206574u;  // "expand 32-byte k"
  uint32_t x0 = s0, x1 = s1, x2 = s2, x3 = s3;
  uint32_t x4 = key.k[0], x5 = key.k[1], x6 = key.k[2], x7 = key.k[3], x8 = key.k[4], x9 = key.k[5], x10 = key.k[6], x11 = key.k[7];
  uint32_t x12 = i0, x13 = i1, x14 = i2, x15 = i3;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int r = 0; r < 10; r++) {
    HIPBFV_QR(x0, x4, x8, x12)
    HIPBFV_QR(x1, x5, x9, x13)
    HIPBFV_QR(x2, x6, x10, x14)
    HIPBFV_QR(x3, x7, x11, x15)
    HIPBFV_QR(x0, x5, x10, x15)
    HIPBFV_QR(x1, x6, x11, x12)
    HIPBFV_QR(x2, x7, x8, x13)
    HIPBFV_QR(x3, x4, x9, x14)
  }
  const uint32_t w[16] = {x0 + s0,        x1 + s1,        x2 + s2,         x3 + s3,         x4 + key.k[0],  x5 + key.k[1],
                          x6 + key.k[2],  x7 + key.k[3],  x8 + key.k[4],   x9 + key.k[5],   x10 + key.k[6], x11 + key.k[7],
                          x12 + i0,       x13 + i1,       x14 + i2,        x15 + i3};
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int j = 0; j < NOUT; j++) out[j] = w[j];
}
#undef HIPBFV_QR

// ---- host side: seeds ----
// 512-bit seed -> the two working keys (see the header comment).  `seed64` = 64 bytes.
inline RngSeed rng_seed_from_512(const void* seed64) {
  RngKey lo, hi;
  std::memcpy(lo.k, seed64, 32);
  std::memcpy(hi.k, static_cast<const uint8_t*>(seed64) + 32, 32);
  RngSeed s;
  for (uint32_t dom = 1; dom <= 2; dom++) {
    uint32_t a[16], b[16];
    chacha20_block<16>(lo, dom, 0x68697062u, 0x66762d6bu, 0x64663031u, a);  // "hipbfv-kdf01"
    chacha20_block<16>(hi, dom, 0x68697062u, 0x66762d6bu, 0x64663031u, b);
    RngKey& dst = dom == 1 ? s.secret : s.pub;
    for (int j = 0; j < 8; j++) dst.k[j] = a[j] ^ b[8 + j];
  }
  return s;
}

// TEST-ONLY seeds: 64 bits of entropy, expanded to the 512-bit form.  Reproducible keys / ciphertexts for the parity tests
// (hipbfv_*SetSeed, hipbfv_KeyGenerator_CreateSeeded, the `seed` argument of hipbfv_batch_encrypt); never a source of keys
// that protect data -- 2^64 candidates can be searched.
inline RngSeed rng_seed_from_u64_for_tests(uint64_t seed) {
  RngKey k{};
  k.k[0] = (uint32_t)seed;
  k.k[1] = (uint32_t)(seed >> 32);
  k.k[2] = 0x74657374u;  // "test"
  uint32_t w[16];
  chacha20_block<16>(k, 0, 0, 0, 0, w);
  return rng_seed_from_512(w);
}

}  // namespace hipbfv
