// sunscreen_amd/csrc/wire.hpp -- SEAL 4.0 wire format (see wire.cpp).  Host-only code: no device access.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace hipbfv {

// kWireSeeded: a seed-compressed ("compact") object -- SEAL stores the first polynomial and a PRNG seed for the second;
// expanding it means reproducing SEAL's Blake2xb / SHAKE256 stream and rejection sampling bit for bit, which nothing here
// could be checked against: refused with its own message instead of loading as garbage
enum WireStatus : int { kWireOk = 0, kWireBadArg = -1, kWireIo = -2, kWireNoZstd = -3, kWireSeeded = -4 };

struct WireCiphertext {
  uint8_t parms_id[32];
  bool is_ntt = false;
  unsigned long long size = 0, n = 0, k = 0, correction = 1;
  double scale = 1.0;
  std::vector<unsigned long long> data;  // [size][k][n]
};

struct WirePlaintext {
  uint8_t parms_id[32];
  double scale = 1.0;
  std::vector<unsigned long long> coeffs;
};

struct WireKSwitchKeys {
  uint8_t parms_id[32];
  std::vector<std::vector<WireCiphertext>> keys;  // [index][decomposition J] -> (2, K+1, N) NTT-form public keys
};

void blake2b_256(const void* data, size_t len, uint8_t out[32]);
// SEAL EncryptionParameters::compute_parms_id for BFV: BLAKE2b-256([1, n, primes..., t])
void seal_parms_id(unsigned long long n, const unsigned long long* primes, size_t count, unsigned long long t, uint8_t out[32]);
bool wire_zstd_available();

int wire_pack_ciphertext(const uint8_t parms_id[32], bool is_ntt, unsigned long long size, unsigned long long n, unsigned long long k, const unsigned long long* data, int compr,
                         std::vector<uint8_t>* out);
// max_body: ceiling on the (decompressed) object body in bytes, 0 = format-wide default; callers with a context pass what
// that context can legally hold
int wire_unpack_ciphertext(const uint8_t* in, size_t in_size, WireCiphertext* ct, size_t* consumed, size_t max_body = 0);
int wire_pack_plaintext(const uint8_t parms_id[32], const unsigned long long* coeffs, unsigned long long count, int compr, std::vector<uint8_t>* out);
int wire_unpack_plaintext(const uint8_t* in, size_t in_size, WirePlaintext* pt, size_t* consumed, size_t max_body = 0);
// keys[index] = list over the decomposition index J of pointers to u64[2][kk][n]; an empty list = key absent
int wire_pack_kswitch(const uint8_t parms_id[32], unsigned long long n, unsigned long long kk, const std::vector<std::vector<const unsigned long long*>>& keys, int compr,
                      std::vector<uint8_t>* out);
int wire_unpack_kswitch(const uint8_t* in, size_t in_size, WireKSwitchKeys* ks, size_t* consumed, size_t max_body = 0, size_t max_index = 0);

}  // namespace hipbfv
