// sunscreen_amd/csrc/evaluator.hpp -- the GPU batch executor for the Evaluator operations.
//
// Mirrors the operation set of `trait seal_fhe::Evaluator` (seal_fhe/src/evaluator.rs:7-280) with an
// extra leading batch dimension that the reference does not have: every method processes `count`
// independent ciphertexts laid out contiguously in HBM (u64[count][size][K][N]) with one sequence of
// kernel launches, instead of one FFI call per graph node (sunscreen_runtime/src/run.rs:160-341).
// All pointers are device pointers; all work is enqueued on the given HIP stream and is asynchronous.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "context.hpp"
#include "kernels.hpp"

namespace hipbfv {

enum Status : int {
  kOk = 0,
  kInvalidArg = -1,
  kTransparent = -2,
  kNoKey = -3,
  kOutOfMemory = -4,
  kHipError = -5,
  kUnsupported = -6,
};

// Stream-aware caching allocator for kernel scratch space.
class ScratchPool {
 public:
  ~ScratchPool();
  void* acquire(size_t bytes, hipStream_t s);
  void release(void* p, hipStream_t s);
  void trim();

 private:
  struct Block {
    void* ptr;
    size_t bytes;
    hipStream_t last;
    hipEvent_t ev;
    bool busy;
  };
  std::mutex mu_;
  std::vector<Block> blocks_;
};

// Per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline figures).
enum KernelId : int {
  kKernNttFwd = 0,
  kKernNttInv,
  kKernBehzExtend,
  kKernTensor,
  kKernBehzFloorSk,
  kKernKsDecompose,
  kKernKsMac,
  kKernKsModdown,
  kKernGalois,
  kKernEltwise,
  kKernPlain,
  kKernKsHead,
  kKernKsMid,
  kKernKsTail,
  kKernMulHead,
  kKernMulMid,
  kKernMulTail,
  kKernCount
};
const char* kernel_name(int id);

class Profiler {
 public:
  ~Profiler();
  bool enabled = false;
  // record the start/stop events around one launch; `units` = work items of that launch (e.g. residue polys)
  void begin(int id, size_t units, hipStream_t s);
  void end(hipStream_t s);
  // synchronise, accumulate finished records into the totals and recycle their events
  void collect();
  void reset();
  double total_ms[kKernCount] = {};
  unsigned long long launches[kKernCount] = {};
  unsigned long long units[kKernCount] = {};

 private:
  struct Rec {
    int id;
    size_t units;
    hipEvent_t a, b;
  };
  std::mutex mu_;
  std::vector<Rec> recs_;
  std::vector<hipEvent_t> free_;
  hipEvent_t get_event();
};

// Pinned host staging blocks for small tables that a launch sequence reads after the call has returned (the per-item key maps).
// Like ScratchPool, but a block is handed out again only after the copy out of it has finished (the HOST overwrites it).
class PinnedPool {
 public:
  ~PinnedPool();
  void* acquire(size_t bytes);
  void release(void* p, hipStream_t s);  // the copies out of the block were enqueued on `s`

 private:
  struct Block {
    void* ptr;
    size_t bytes;
    hipEvent_t ev;
    bool busy, pending;
  };
  std::mutex mu_;
  std::vector<Block> blocks_;
};

// Which key-switching key each item of a batch uses.  The reference hands the keys over with every call
// (sunscreen_runtime/src/run.rs:100-105: `relin_keys: &Option<&RelinearizationKeys>`, `galois_keys`; runtime.rs:310-327), so a
// server that batches the calls of many clients holds one key set per client: item i of the batch uses
// keys[index[(first + i) % period]] (period = the number of input sets: the graph executor's merged launches number their
// ciphertexts member-major, item = member * period + set).  keys == nullptr: the one key `key` for every item.
struct KeySel {
  const u64* key = nullptr;          // device, u64[K][2][K+1][N]
  const u64* const* keys = nullptr;  // HOST array of nkeys device pointers
  u32 nkeys = 0;
  const u32* index = nullptr;        // HOST array of `period` key indices
  size_t period = 0;
  size_t first = 0;                  // the batch's item 0 is item `first` of the selection
  KeySel() = default;
  KeySel(const u64* one) : key(one) {}  // NOLINT: every single-key call site passes its pointer
  bool per_item() const { return keys != nullptr; }
  bool present() const { return per_item() ? nkeys != 0 : key != nullptr; }
};

// RAII lease of a scratch-pool buffer on a stream
struct ScratchGuard {
  ScratchPool& pool;
  hipStream_t s;
  void* p;
  ScratchGuard(ScratchPool& pl, size_t bytes, hipStream_t st) : pool(pl), s(st), p(pl.acquire(bytes, st)) {}
  ~ScratchGuard() {
    if (p) pool.release(p, s);
  }
  ScratchGuard(const ScratchGuard&) = delete;
  ScratchGuard& operator=(const ScratchGuard&) = delete;
};

// Transparent-result watch of the batched path.  While a WatchScope is alive on the calling thread, every Evaluator
// operation that produces ciphertexts records, in the device word it was given, the smallest batch index whose result is
// transparent (all polynomials but the first are zero; 0xFFFFFFFF = none): the reference's SEAL build throws on such a
// result (seal_fhe/build.rs:46-66, sunscreen/tests/features.rs:8-34).  The handle-level C entry points do their own
// per-call check and never open a scope.
struct WatchScope {
  explicit WatchScope(u32* status_dev);
  ~WatchScope();
  u32* prev;
};

class Evaluator {
 public:
  explicit Evaluator(Context* ctx);
  ~Evaluator();
  // the evaluator's own status word for hipbfv_batch_* callers (nullptr if the allocation failed); read-and-reset:
  // *first_bad = smallest transparent item since the last call, 0xFFFFFFFF if none.  Synchronises `s`.
  u32* batch_status() const { return status_dev_; }
  int take_status(u32* status_dev, u32* first_bad, hipStream_t s);
  int note_result(const u64* ct, u32 size, u32 residues, size_t count, hipStream_t s);
  static u32* watch_status();  // the status word of the calling thread's innermost WatchScope (nullptr outside one)
  bool few_for_split_mul(size_t count) const;
  bool few_for_split_ks(size_t count) const;
  bool few_for_fused(size_t count) const;
  Profiler& profiler() { return prof_; }
  ScratchPool& scratch() { return pool_; }
  Context* ctx() const { return ctx_; }

  // ---- SURVEY 8a rows a1-a5, batched ----
  int multiply(const u64* a, u32 sa, const u64* b, u32 sb, u64* out, size_t count, hipStream_t s, bool watch = true);
  // addend (optional, the three key-switching operations): ciphertexts u64[count][2][K][N] added to the results inside the last kernel
  // rk / key: one key for the whole batch (a device pointer converts) or a per-item selection (KeySel above)
  int relinearize(const u64* ct3, const KeySel& rk, u64* out2, size_t count, hipStream_t s, const u64* addend = nullptr, bool watch = true);
  // members / per (program executor, merged launches): a DEVICE table of per-member epilogues (kernels.hpp MemberTail) -- item i belongs to
  // member i / per and is written to that member's own buffer as mult * product + sign * extra; out2 and addend are unused then, and
  // the caller notes the results itself.  Only where member_tail_ok(count) says so (the all-FP64 fused path).
  // heads (with members): the members' operands where they are (kernels.hpp MemberHead) -- a and b are unused then, `square` says whether every
  // member multiplies a ciphertext by itself
  int multiply_relin(const u64* a, const u64* b, const KeySel& rk, u64* out2, size_t count, hipStream_t s, const u64* addend = nullptr,
                     const MemberTail* members = nullptr, u32 per = 0, const MemberHead* heads = nullptr, bool heads_square = false);
  bool member_tail_ok(size_t count) const;
  // out2 = (sigma_g(c0), 0) + switch_key(sigma_g(c1), key)
  int apply_galois(const u64* ct2, u32 galois_elt, const KeySel& key, u64* out2, size_t count, hipStream_t s, const u64* addend = nullptr);
  int mod_switch_next(const u64* ct, u32 size, u64* out, size_t count, hipStream_t s);  // out has K-1 residues per polynomial
  int add(const u64* a, const u64* b, u64* out, u32 size, size_t count, hipStream_t s);
  int sub(const u64* a, const u64* b, u64* out, u32 size, size_t count, hipStream_t s);
  int negate(const u64* a, u64* out, u32 size, size_t count, hipStream_t s);
  // plain: u64[count][N] (pstride = N) or one shared plaintext (pstride = 0), coefficients < t, zero padded
  int add_plain(const u64* ct, u32 size, const u64* plain, size_t pstride, u64* out, size_t count, hipStream_t s);
  int sub_plain(const u64* ct, u32 size, const u64* plain, size_t pstride, u64* out, size_t count, hipStream_t s);
  int multiply_plain(const u64* ct, u32 size, const u64* plain, size_t pstride, u64* out, size_t count, hipStream_t s);
  // monomial fast path of SEAL multiply_plain_normal: plaintext = coeff * x^exponent
  int multiply_plain_mono(const u64* ct, u32 size, u64 coeff, u32 exponent, u64* out, size_t count, hipStream_t s);
  // flags[i] = 1 if ciphertext i is NOT transparent (some word of polys 1.. is non-zero); flags must be zeroed
  int nonzero_tail(const u64* ct, u32 size, u32* flags, size_t count, hipStream_t s);

  // ---- the steps either side of the path (evaluator_client.cpp) ----
  int batch_encode(const u64* values, u64* plain, size_t count, bool is_signed, u32* bad_host, hipStream_t s);
  int batch_decode(const u64* plain, u64* values, size_t count, bool is_signed, hipStream_t s);
  int decrypt(const u64* ct, u32 size, const u64* sk_ntt, u64* plain, size_t count, hipStream_t s);
  int keygen_secret(const RngSeed& seed, u64* sk_coeff, u64* sk_ntt, hipStream_t s);
  int keygen_zero_encryptions(const RngSeed& seed, u64 stream, const u64* sk_ntt, const u64* w, u64* key, u32 count, hipStream_t s);
  // single encryptions that also return the sampled polynomials (fork-only API used by logproof), and secret-key encryption
  int encrypt_components(const u64* plain, const u64* pk, const RngSeed& seed, u64 op, bool no_special, u64* ct2, u64* u_out, u64* e_out, hipStream_t s);
  int encrypt_symmetric(const u64* plain, const u64* sk_ntt, const RngSeed& seed, u64 stream, u64* ct2, u64* e_out, hipStream_t s);
  int key_to_coeff(const u64* key, u32 polys, u64* out, hipStream_t s);
  int crt_compose(const u64* consts, u32 kc, const u64* in, u64* out, u32 polys, hipStream_t s);
  int crt_decompose(u32 kc, const u64* in, u64* out, u32 polys, hipStream_t s);
  int keygen_kswitch(const RngSeed& seed, u64 stream, const u64* sk_coeff, const u64* sk_ntt, u32 galois_elt, u64* key, hipStream_t s);
  // watch_zero: an all-zero plaintext raises the transparent-result status of the enclosing WatchScope -- 1: the plaintexts are
  // the items of a batch (item = index), 2: they are shared by the whole batch (item 0): the graph executor's transform-domain
  // sums never form the single product SEAL's multiply_plain would have refused
  int plain_to_ntt(const u64* plain, size_t pstride, u64* pntt, size_t count, hipStream_t s, int watch_zero = 0);
  int ct_to_ntt(const u64* ct, u32 size, u64* ctn, size_t count, hipStream_t s);
  int dot_plain_ntt(const u64* ctn, u32 cols, const u64* pntt, u32 rows, u64* out, hipStream_t s);
  // the graph executor's form: ctn u64[cols][batch][2][K][N] (transform domain), tab [rows][cols] device descriptors of
  // transform-domain plaintexts, out u64[rows][batch][2][K][N] in coefficient form
  int dot_plain_tab(const u64* ctn, u32 cols, const PlainNttRef* tab, u32 rows, u32 batch, u64* out, hipStream_t s);
  // ct (.) plaintext given in transform form (u64[K][N] at pntt + item * pnstride): transform, product, inverse transform
  int multiply_plain_ntt(const u64* ct, u32 size, const u64* pntt, size_t pnstride, u64* out, size_t count, hipStream_t s);
  // the transparent-result watch of one n-ary sum launch (device descriptor table)
  int note_nary(const NaryOut* douts, u32 nouts, u32 batch, hipStream_t s);
  int phase(const u64* ct, u32 size, const u64* sk_ntt, u64* out, size_t count, hipStream_t s);
  int encrypt(const u64* plain, size_t pstride, const u64* pk, const RngSeed& seed, u64 first_op, u64* ct2, size_t count, hipStream_t s);

  // ---- NTT entry points (BASELINE config 2) ----
  // data: u64[polys][N]; polynomial p uses key-level prime (p % nprimes)
  int ntt(u64* data, size_t polys, u32 nprimes, bool inverse, hipStream_t s);

  u32 galois_elt_from_step(int step) const;  // 0 if |step| >= n/2
  size_t chunk_ops() const { return chunk_ops_; }
  void set_chunk_ops(size_t c) { chunk_ops_ = c ? c : 1; }

 private:
  int key_switch(const u64* target, size_t tstride, const u64* key, const u64* base, size_t bstride, u32 base_mask, u64* out2,
                 size_t count, u64* scratch, hipStream_t s, const u64* extra = nullptr, KeyMap km = KeyMap{}, u32 ginv = 0);
  bool ks_split_for(size_t count) const;  // key_switch takes the head / middle / tail kernels for a batch of this size
  size_t ks_scratch_words() const;
  // the device tables of a per-item key selection for one call: `order` holds, for every chunk of `chunk` items, the chunk's items
  // (numbered from 0 within the chunk) sorted by key -- the launch over chunk c reads order + c * chunk
  struct KeyMapLease {
    Evaluator* ev = nullptr;
    hipStream_t s = nullptr;
    void* dev = nullptr;
    void* host = nullptr;
    KeyMap km;
    KeyMap at(size_t off) const { return km.keys ? KeyMap{km.keys, km.order + off} : KeyMap{}; }
    ~KeyMapLease();
  };
  int stage_keymap(const KeySel& sel, size_t count, size_t chunk, hipStream_t s, KeyMapLease& lease);
  PinnedPool pinned_;
  Context* ctx_;
  u32* status_dev_ = nullptr;
  ScratchPool pool_;
  Profiler prof_;
  size_t chunk_ops_;
  bool split_ks_ = true;   // head / middle / tail split transforms for key switching (kernels_split.hip)
  bool split_mul_ = true;  // ... and for the BEHZ multiply
  bool fuse_head_ = true;      // ... and c2 formed inside the key switch's first kernel
  bool small_batch_ = true;    // a few ciphertexts take the whole-polynomial pipelines (HIPBFV_NO_SMALL_BATCH=1: pipelines chosen by parameters only)
  bool fuse_galois_ = true;    // rotations: the automorphism read through the key switch's head / tail loads (no rotated copy)
  bool fuse_mulrelin_ = true;  // multiply_relin: c0, c1 of the product formed inside the key switch's last kernel
};

}  // namespace hipbfv
