// sunscreen_amd/csrc/context.hpp -- host-side BFV context: parameter validation, prime/root
// search and the precomputed tables that are uploaded once to HBM.
//
// Replaces what SEALContext_Create builds inside SEAL for the reference
// (seal_fhe/src/context.rs:63-80 -> bindgen::SEALContext_Create): NTTTables per prime and the
// RNSTool constants of the first data level.  Written from the published algorithms
// (Harvey NTT tables, BEHZ base conversion, hybrid key switching); shares no code with oracle/.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "devctx.hpp"

namespace hipbfv {

bool is_prime_u64(u64 v);
// Host-only view of the FP64 range plans of a prime at degree 2^logn (no device needed: tests/test_fp64_range_plan_cpu.py):
// out = {use_f64, fwd_reduce_mask, inv_reduce_mask, split_ok, split_fwd_mask, split_inv_mask}
void debug_f64_plan(u64 q, int logn, u32 out[6]);
// primes == 1 (mod factor), descending from just below 2^bits
std::vector<u64> find_primes(u64 factor, int bits, size_t count);
u64 minimal_primitive_root(u64 two_n, u64 q);
std::vector<u64> default_coeff_modulus(u64 n, int sec_level);
int max_coeff_bit_count(u64 n, int sec_level);

class Context {
 public:
  // key_primes: data-level primes followed by the special prime (a single prime = no key switching)
  static Context* create(u32 n, const std::vector<u64>& key_primes, u64 plain_modulus, int device, std::string* err);
  ~Context();

  u32 n() const { return host_.n; }
  u32 K() const { return host_.K; }
  u32 KK() const { return host_.KK; }
  u32 S() const { return host_.S; }
  u64 t() const { return host_.t; }
  int device() const { return device_; }
  const DevCtx& host() const { return host_; }
  const DevCtx* dev() const { return dev_; }
  const std::vector<u64>& key_primes() const { return key_primes_; }
  bool batching() const { return batching_; }
  // Modulus-switching chain (SEAL context_data->next_context_data()): the context of the next level drops the last
  // data prime and keeps the special prime; created on first use; nullptr (and *err) at the end of the chain.
  std::shared_ptr<Context> next_level(std::string* err);
  std::shared_ptr<Context> peek_next() {  // the next level if it has been created already
    std::lock_guard<std::mutex> g(next_mu_);
    return next_;
  }
  int level() const { return level_; }
  void set_chain_enabled(bool on) { chain_enabled_ = on; }  // SEALContext_Create(expand_mod_chain = false)  // 0 = the context the user created
  const u32* batch_index_map() const { return batch_map_; }  // device u32[n]: BatchEncoder matrix_reps_index_map
  size_t ct_words(size_t size) const { return size * (size_t)host_.K * host_.n; }
  size_t key_words() const { return (size_t)host_.K * 2 * host_.KK * host_.n; }

 private:
  Context() = default;
  DevCtx host_{};
  DevCtx* dev_ = nullptr;
  MulOp* tw_fwd_ = nullptr;
  MulOp* tw_inv_ = nullptr;
  u32* batch_map_ = nullptr;
  int device_ = 0;
  int level_ = 0;
  bool chain_enabled_ = true;
  std::mutex next_mu_;
  std::shared_ptr<Context> next_;
  bool batching_ = false;
  std::vector<u64> key_primes_;
};

}  // namespace hipbfv
