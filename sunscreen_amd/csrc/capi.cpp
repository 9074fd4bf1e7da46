// sunscreen_amd/csrc/capi.cpp -- the C ABI declared in include/hipbfv.h.
//
// Part 1 implements the SEAL C entry points that seal_fhe binds for the evaluator path (names,
// arity and HRESULT behaviour as used in seal_fhe/src/evaluator_base.rs:55-407,
// bfv_evaluator.rs:12-248, plaintext_ciphertext.rs:36-504, context.rs:63-115, modulus.rs,
// encryption_parameters.rs); Part 2 the batched device-pointer extension.
#include "../../include/hipbfv.h"

#include <hip/hip_runtime.h>
#include <sys/random.h>

#include <cerrno>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "evaluator.hpp"
#include "flat_combiner.hpp"
#include "program.hpp"
#include "wire.hpp"

using namespace hipbfv;

namespace {

thread_local std::string tls_error;
// One device per process (one process per GPU): hipbfv_set_device() picks it before the first context
// exists.  Every C entry point makes it the calling thread's current device first (enter_thread), so host threads that never
// called hipSetDevice themselves -- the rayon workers of sunscreen_runtime/src/run.rs:415-469 -- allocate, create their
// stream and launch on the device the contexts, twiddles and keys live on.
std::atomic<int> g_device{0};
std::atomic<long> g_live_contexts{0};
bool g_throw_transparent = true;

void enter_thread() {
  thread_local int tls_device = -1;
  const int want = g_device.load(std::memory_order_relaxed);
  if (tls_device != want) {
    if (hipSetDevice(want) != hipSuccess) (void)hipGetLastError();  // no GPU: the host-only entry points still work
    tls_device = want;
  }
}

long fail(long hr, const char* msg) {
  tls_error = msg ? msg : "";
  return hr;
}

long from_status(int st) {
  switch (st) {
    case kOk: return HIPBFV_S_OK;
    case kInvalidArg: return fail(HIPBFV_E_INVALIDARG, "invalid argument");
    case kTransparent: return fail(HIPBFV_COR_E_INVALIDOPERATION, "result ciphertext is transparent");
    case kNoKey: return fail(HIPBFV_E_INVALIDARG, "required key-switching key is not present");
    case kOutOfMemory: return fail(HIPBFV_E_OUTOFMEMORY, "out of device memory");
    case kUnsupported: return fail(HIPBFV_COR_E_INVALIDOPERATION, "unsupported parameters for the HIP path");
    default: {
      static thread_local char buf[160];
      snprintf(buf, sizeof(buf), "HIP runtime error: %s", hipGetErrorString(hipGetLastError()));
      return fail(HIPBFV_E_UNEXPECTED, buf);
    }
  }
}

enum Magic : uint32_t {
  kMagicModulus = 0x4D4F4431,
  kMagicParams = 0x50524D31,
  kMagicContext = 0x43545831,
  kMagicPlain = 0x504C4E31,
  kMagicCipher = 0x43504831,
  kMagicKeys = 0x4B535731,
  kMagicEval = 0x45564C31,
  kMagicProgram = 0x50524731,
  kMagicSecretKey = 0x534B4531,
  kMagicPublicKey = 0x504B4531,
  kMagicEncoder = 0x42454E31,
  kMagicDecryptor = 0x44454331,
  kMagicEncryptor = 0x454E4331,
  kMagicKeyGen = 0x4B47454E,
  kMagicPolyArray = 0x504F4C59,
};

struct Obj {
  uint32_t magic;
  explicit Obj(uint32_t m) : magic(m) {}
  virtual ~Obj() { magic = 0; }
};

template <typename T>
T* as(void* p, uint32_t magic) {
  if (!p) return nullptr;
  Obj* o = static_cast<Obj*>(p);
  return o->magic == magic ? static_cast<T*>(o) : nullptr;
}

struct ModulusObj : Obj {
  u64 value;
  explicit ModulusObj(u64 v) : Obj(kMagicModulus), value(v) {}
};

struct ParamsObj : Obj {
  uint8_t scheme = 1;
  u64 n = 0;
  std::vector<u64> coeff;
  u64 plain = 0;
  ParamsObj() : Obj(kMagicParams) {}
};

struct ContextObj : Obj {
  std::shared_ptr<Context> ctx;
  ContextObj() : Obj(kMagicContext) {}
};

struct PlainObj : Obj {
  std::vector<u64> coeffs;
  PlainObj() : Obj(kMagicPlain) {}
};

// Device buffers for ciphertexts are recycled: hipFree synchronises the whole device, which would
// serialise concurrent evaluator threads (sunscreen_runtime/src/run.rs:415-469 calls from a rayon pool).
class BufferCache {
 public:
  BufferCache() {
    // bytes the cache may hold on to (freed blocks beyond it go back to the device at once): long-lived processes that
    // create and drop key sets (hundreds of MB each) or many contexts do not grow without bound
    if (const char* e = std::getenv("HIPBFV_CACHE_BYTES")) budget_ = std::strtoull(e, nullptr, 10);
  }
  u64* get(size_t words) {
    {
      std::lock_guard<std::mutex> g(mu_);
      auto it = free_.find(words);
      if (it != free_.end() && !it->second.empty()) {
        u64* p = it->second.back();
        it->second.pop_back();
        cached_ -= words * sizeof(u64);
        return p;
      }
    }
    void* p = nullptr;
    if (hipMalloc(&p, words * sizeof(u64)) != hipSuccess) {
      (void)hipGetLastError();
      drain();
      if (hipMalloc(&p, words * sizeof(u64)) != hipSuccess) return nullptr;
    }
    return (u64*)p;
  }
  void put(u64* p, size_t words) {
    if (!p) return;
    const size_t bytes = words * sizeof(u64);
    {
      std::lock_guard<std::mutex> g(mu_);
      if (cached_ + bytes <= budget_) {
        free_[words].push_back(p);
        cached_ += bytes;
        return;
      }
    }
    (void)hipFree(p);  // over budget: synchronises the device, which only happens for blocks the cache refuses
  }
  void drain() {
    std::lock_guard<std::mutex> g(mu_);
    for (auto& kv : free_)
      for (u64* p : kv.second) (void)hipFree(p);
    free_.clear();
    cached_ = 0;
  }
  size_t cached_bytes() {
    std::lock_guard<std::mutex> g(mu_);
    return cached_;
  }

 private:
  std::mutex mu_;
  std::map<size_t, std::vector<u64*>> free_;
  size_t cached_ = 0;
  size_t budget_ = (size_t)4 << 30;
};
BufferCache g_buffers;

struct CipherObj : Obj {
  std::shared_ptr<Context> ctx;
  u32 size = 0;
  u64* dev = nullptr;
  size_t words = 0;
  std::vector<u64> host;  // lazily filled host mirror for GetDataAt
  std::mutex host_mu;
  bool host_valid = false;
  CipherObj() : Obj(kMagicCipher) {}
  ~CipherObj() override { g_buffers.put(dev, words); }
  void adopt(const std::shared_ptr<Context>& c, u32 sz, u64* buf, size_t w) {
    if (dev && dev != buf) g_buffers.put(dev, words);
    ctx = c;
    size = sz;
    dev = buf;
    words = w;
    host_valid = false;
  }
};

struct KeysObj : Obj {
  std::shared_ptr<Context> ctx;
  std::map<u32, u64*> keys;  // index -> device key u64[K][2][K+1][N]
  // keys re-packed for lower levels of the modulus-switching chain: (level context, index) -> u64[K'][2][K'+1][N]
  std::map<std::pair<Context*, u32>, std::pair<u64*, size_t>> lower;
  std::mutex mu;
  KeysObj() : Obj(kMagicKeys) {}
  ~KeysObj() override {
    for (auto& kv : keys) g_buffers.put(kv.second, ctx ? ctx->key_words() : 0);
    for (auto& kv : lower) g_buffers.put(kv.second.first, kv.second.second);
  }
  const u64* find(u32 index) const {
    auto it = keys.find(index);
    return it == keys.end() ? nullptr : it->second;
  }
};

// ---- concurrent handle-level calls combined into batched launches ----
// sunscreen_runtime runs ready graph nodes from a rayon pool (run.rs:415-469): many host threads, each asking ONE evaluator
// for one multiply / relinearize / rotation at a time.  Issued separately, such calls overlap poorly on the device (one
// operation's kernels occupy a few CUs for ~100 us; the runtime multiplexes the threads' streams onto a handful of hardware
// queues: 5 -> 8-12 K multiply+relinearize/s from 1 -> 8 threads, fewer with 32).  Flat combining instead: a caller
// that finds no operation of its evaluator in flight runs its own at once, exactly as before; callers arriving while one is in
// flight queue up, and whoever finds the evaluator free next takes every queued request of one kind (same Galois element; since
// r06 the keys may differ: many clients, each with its own keys, on one evaluator) and runs them as ONE batch through the batched kernels -- operands gathered into a staging batch and results
// scattered to the callers' buffers by two copy kernels that read a pointer table in pinned host memory, one transparent flag
// per item, one stream synchronisation for all.  Batch items are independent: the bits do not depend on who was combined with whom.
struct CombReq {
  // 0 multiply (2 x 2 -> 3 polynomials), 1 relinearize (3 -> 2), 2 Galois automorphism + key switch (2 -> 2), 3 add, 4 sub (2, 2 -> 2),
  // 5 add_plain, 6 sub_plain, 7 multiply_plain (2 polynomials and a plaintext of N coefficients in in1 -> 2)
  int kind = 0;
  const u64* in0 = nullptr;
  const u64* in1 = nullptr;
  const u64* key = nullptr;
  u32 elt = 0;
  u64 mono_coeff = 0;  // multiply_plain by a monomial (SEAL's shortcut): coefficient and exponent, used when the request runs alone
  u32 mono_exp = 0;
  u64* out = nullptr;
  int status = 0;
  int nonzero = -1;  // transparent check of the result: 1 / 0 (made where the request ran, together with the synchronisation)
  std::atomic<bool> done{false};
};
using Combiner = FlatCombiner<CombReq>;  // flat_combiner.hpp: the queueing protocol (ThreadSanitizer-tested on its own)
#ifndef HIPBFV_COMBINE_LEADERS
#define HIPBFV_COMBINE_LEADERS 1
#endif
constexpr int kCombineLeaders = HIPBFV_COMBINE_LEADERS;
constexpr size_t kCombineMax = 64;  // HIPBFV_NO_COMBINE=1 (read at the first call): every handle-level call launches on its own

struct EvalObj : Obj {
  std::shared_ptr<Context> ctx;
  std::unique_ptr<Evaluator> ev;
  Combiner comb;
  // evaluators of the lower levels of the modulus-switching chain (SEAL's single Evaluator serves every level of
  // its context; here every level has its own tables): created on first use
  std::mutex mu;
  std::map<Context*, std::unique_ptr<EvalObj>> lower;
  bool batch_watch = true;  // hipbfv_batch_*: record transparent results in the evaluator's status word
  EvalObj() : Obj(kMagicEval) {}
};

// is `c` the context `top` or one of the (already created) levels below it?
bool in_chain(const std::shared_ptr<Context>& top, const Context* c) {
  for (std::shared_ptr<Context> p = top; p; p = p->peek_next())
    if (p.get() == c) return true;
  return false;
}

// the evaluator serving ciphertexts of level context `c` (nullptr if c does not belong to top's chain)
EvalObj* level_eval(EvalObj* top, const std::shared_ptr<Context>& c) {
  if (!top || !c) return nullptr;
  if (top->ctx.get() == c.get()) return top;
  if (!in_chain(top->ctx, c.get())) return nullptr;
  std::lock_guard<std::mutex> g(top->mu);
  auto& slot = top->lower[c.get()];
  if (!slot) {
    slot.reset(new EvalObj());
    slot->ctx = c;
    slot->ev.reset(new Evaluator(c.get()));
  }
  return slot.get();
}

// SecretKey: u64[KK][N] NTT form (SEAL SecretKey data); PublicKey: u64[2][KK][N] NTT form.  The device buffer is
// shared with every Decryptor / Encryptor created from the handle.
struct KeyBuffer {
  std::shared_ptr<Context> ctx;
  u64* dev = nullptr;
  size_t words = 0;
  ~KeyBuffer() { g_buffers.put(dev, words); }
};
struct AsymKeyObj : Obj {
  std::shared_ptr<KeyBuffer> key;
  explicit AsymKeyObj(uint32_t magic) : Obj(magic) {}
};

struct EncoderObj : Obj {
  std::shared_ptr<Context> ctx;
  std::unique_ptr<Evaluator> ev;
  EncoderObj() : Obj(kMagicEncoder) {}
};
struct DecryptorObj : Obj {
  std::shared_ptr<Context> ctx;
  EvalObj core;  // evaluators per level of the modulus-switching chain (ciphertexts of every level decrypt)
  std::shared_ptr<KeyBuffer> sk;
  DecryptorObj() : Obj(kMagicDecryptor) {}
};
struct EncryptorObj : Obj {
  std::shared_ptr<Context> ctx;
  std::unique_ptr<Evaluator> ev;
  std::shared_ptr<KeyBuffer> pk;
  std::shared_ptr<KeyBuffer> sk;  // optional: symmetric encryption
  RngSeed seed{};  // 512 bits of OS entropy per Encryptor (rng.hpp); hipbfv_Encryptor_SetSeed replaces it for tests
  std::mutex mu;
  u64 next_op = 0;  // block counter: every encryption of one Encryptor uses fresh randomness
  EncryptorObj() : Obj(kMagicEncryptor) {}
};

struct KeyGenObj : Obj {
  std::shared_ptr<Context> ctx;
  std::unique_ptr<Evaluator> ev;
  std::shared_ptr<KeyBuffer> sk;        // NTT form, what SecretKey handles share
  std::shared_ptr<KeyBuffer> sk_coeff;  // the ternary polynomial in coefficient form (Galois keys permute it)
  RngSeed seed{};  // 512 bits of OS entropy per KeyGenerator (rng.hpp)
  std::mutex mu;
  u64 next_stream = 1;  // every generated key draws from its own stream (block-input word)
  KeyGenObj() : Obj(kMagicKeyGen) {}
};

// The fork's PolynomialArray (seal_fhe/src/data_structures.rs:17-304): `polys` polynomials over the first `kc` primes of
// the context, coefficient form, either RNS u64[polys][kc][N] or multiprecision u64[polys][N][kc] (little-endian limbs)
struct PolyArrayObj : Obj {
  std::shared_ptr<Context> ctx;
  u32 polys = 0, kc = 0;
  u64* dev = nullptr;
  size_t words = 0;
  bool rns = true, reserved = false;
  PolyArrayObj() : Obj(kMagicPolyArray) {}
  ~PolyArrayObj() override { g_buffers.put(dev, words); }
};

struct ProgramObj : Obj {
  Program prog;
  ProgramObj() : Obj(kMagicProgram) {}
};

// 512 bits from the kernel's CSPRNG (getrandom(2)); keys are never derived from anything weaker: no entropy -> the
// operation fails (E_UNEXPECTED), there is no fallback seed
bool os_seed(RngSeed* out) {
  uint8_t raw[64];
  size_t got = 0;
  while (got < sizeof(raw)) {
    const ssize_t r = getrandom(raw + got, sizeof(raw) - got, 0);
    if (r < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    got += (size_t)r;
  }
  *out = rng_seed_from_512(raw);
  std::memset(raw, 0, sizeof(raw));
  return true;
}

// one non-blocking stream per host thread: concurrent handle-level calls do not serialise on the null stream
hipStream_t thread_stream() {
  // a stream belongs to the device that was current when it was created: cache one per (thread, device)
  thread_local hipStream_t streams[16] = {};
  const int dev = g_device.load(std::memory_order_relaxed) & 15;
  if (!streams[dev]) (void)hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking);
  return streams[dev];
}

long sync_stream(hipStream_t s) {
  if (hipStreamSynchronize(s) != hipSuccess) return from_status(kHipError);
  return HIPBFV_S_OK;
}

// Device-to-device copy that is complete when it returns.  A plain hipMemcpy(D2D) is only ordered on
// the null stream, which the per-thread non-blocking streams do not synchronise with.
hipError_t copy_d2d(void* dst, const void* src, size_t bytes) {
  hipStream_t s = thread_stream();
  hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(s);
}

// The key-switching key `index` for the level e serves: the handle's own buffer at its own level, a re-packed copy
// (digits J < K', residues {0..K'-1, special}) below it.  nullptr if absent or foreign.
const u64* level_key(KeysObj* k, EvalObj* e, u32 index) {
  if (!k || !k->ctx) return nullptr;
  const u64* top = k->find(index);
  if (!top) return nullptr;
  if (k->ctx.get() == e->ctx.get()) return top;
  if (!in_chain(k->ctx, e->ctx.get())) return nullptr;
  std::lock_guard<std::mutex> g(k->mu);
  auto it = k->lower.find({e->ctx.get(), index});
  if (it != k->lower.end()) return it->second.first;
  const size_t n = k->ctx->n(), KKt = k->ctx->KK(), Kl = e->ctx->K(), KKl = e->ctx->KK();
  const size_t words = e->ctx->key_words();
  u64* buf = g_buffers.get(words);
  if (!buf) return nullptr;
  for (size_t J = 0; J < Kl; J++)
    for (size_t c = 0; c < 2; c++)
      for (size_t I = 0; I < KKl; I++) {
        const size_t It = I < Kl ? I : KKt - 1;  // the special prime is the last row at every level
        if (copy_d2d(buf + ((J * 2 + c) * KKl + I) * n, top + ((J * 2 + c) * KKt + It) * n, n * sizeof(u64)) != hipSuccess) {
          g_buffers.put(buf, words);
          return nullptr;
        }
      }
  k->lower[{e->ctx.get(), index}] = {buf, words};
  return buf;
}

bool same_context(const CipherObj* a, const EvalObj* e) { return a->ctx && a->ctx.get() == e->ctx.get() && a->dev && a->size >= 2; }

// transparent check of the finished result (SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT); stream already holds the op
// The pointer / flag table of a combined batch: pinned host memory the device addresses directly, one per host thread
// (only the thread that runs a batch uses its own).
struct CombTable {
  const u64* src[2 * kCombineMax];
  u64* dst[kCombineMax];
  u32 flags[kCombineMax];
};

// Runs `batch` (>= 1 requests of one kind / key / element) on stream s and fills in status (and nonzero) of each.
// Every request is complete on the device and checked when this returns -- also a lone one: its owner may be another thread,
// whose stream knows nothing about this one, and holding the evaluator until the operation has finished is what lets the
// requests that arrive meanwhile pile up into the next batch (with the lone request merely launched, four threads ran
// four interleaved single operations: 10 K instead of 27 K multiply+relinearize/s).
void combine_execute(EvalObj* e, const std::vector<CombReq*>& batch, hipStream_t s) {
  Evaluator& ev = *e->ev;
  CombReq& r0 = *batch[0];
  const size_t c = batch.size();
  // Every failure path drains the stream first: kernels of this batch may already be queued on it, and the requests' owners
  // recycle their buffers (g_buffers) the moment they see a status -- on OTHER threads' streams, which know nothing of this one.
  auto all = [&](int st) {
    if (st != kOk) (void)hipStreamSynchronize(s);
    for (CombReq* r : batch) r->status = st;
  };
  const size_t poly = (size_t)e->ctx->K() * e->ctx->n();
  thread_local CombTable* table = nullptr;
  if (!table && hipHostMalloc((void**)&table, sizeof(CombTable), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
    table = nullptr;
    (void)hipGetLastError();
    return all(kOutOfMemory);
  }
  if (c == 1) {
    r0.status = r0.kind == 0 ? ev.multiply(r0.in0, 2, r0.in1, 2, r0.out, 1, s)
              : r0.kind == 1 ? ev.relinearize(r0.in0, r0.key, r0.out, 1, s)
              : r0.kind == 2 ? ev.apply_galois(r0.in0, r0.elt, r0.key, r0.out, 1, s)
              : r0.kind == 3 ? ev.add(r0.in0, r0.in1, r0.out, 2, 1, s)
              : r0.kind == 4 ? ev.sub(r0.in0, r0.in1, r0.out, 2, 1, s)
              : r0.kind == 5 ? ev.add_plain(r0.in0, 2, r0.in1, 0, r0.out, 1, s)
              : r0.kind == 6 ? ev.sub_plain(r0.in0, 2, r0.in1, 0, r0.out, 1, s)
              : r0.mono_coeff ? ev.multiply_plain_mono(r0.in0, 2, r0.mono_coeff, r0.mono_exp, r0.out, 1, s)
                              : ev.multiply_plain(r0.in0, 2, r0.in1, 0, r0.out, 1, s);
    if (r0.status) {
      (void)hipStreamSynchronize(s);
      return;
    }
    const size_t out_words = (r0.kind == 0 ? 3 : 2) * poly;
    table->flags[0] = 1u;
    if (g_throw_transparent && launch_transparent_flags(r0.out, out_words, poly, table->flags, 1, s) != hipSuccess) return all(kHipError);
    if (hipStreamSynchronize(s) != hipSuccess) return all(kHipError);
    r0.nonzero = table->flags[0] ? 1 : 0;
    return;
  }
  if (r0.kind == 3 || r0.kind == 4) {  // add / sub: one pass over the callers' own buffers, no staging
    for (size_t i = 0; i < c; i++) {
      table->src[i] = batch[i]->in0;
      table->src[c + i] = batch[i]->in1;
      table->dst[i] = batch[i]->out;
      table->flags[i] = 1u;
    }
    if (launch_eltwise_items(e->ctx->dev(), (u32)e->ctx->n(), (u32)e->ctx->K(), table->src, table->src + c, table->dst, r0.kind - 3, c, s) != hipSuccess)
      return all(kHipError);
    if (g_throw_transparent && launch_transparent_flags_items(table->dst, 2 * poly, poly, table->flags, c, s) != hipSuccess) return all(kHipError);
    if (hipStreamSynchronize(s) != hipSuccess) return all(kHipError);
    for (size_t i = 0; i < c; i++) {
      batch[i]->status = kOk;
      batch[i]->nonzero = table->flags[i] ? 1 : 0;
    }
    return;
  }
  bool squares = r0.kind == 0;  // every request multiplies a ciphertext by itself: one staged operand, the squaring kernels
  for (const CombReq* r : batch) squares = squares && r->in0 == r->in1;
  const bool plain_op = r0.kind >= 5;
  const size_t in_polys = r0.kind == 1 ? 3 : 2, nin = r0.kind == 0 && !squares ? 2 : 1, out_polys = r0.kind == 0 ? 3 : 2;
  const size_t in_words = in_polys * poly, out_words = out_polys * poly, pl_words = plain_op ? e->ctx->n() : 0;
  ScratchGuard sg(ev.scratch(), c * (nin * in_words + out_words + pl_words) * sizeof(u64), s);
  if (!sg.p) return all(kOutOfMemory);
  u64* in_stage = (u64*)sg.p;                      // [nin][c][in_polys][K][N]
  u64* out_stage = in_stage + c * nin * in_words;  // [c][out_polys][K][N]
  u64* pl_stage = out_stage + c * out_words;       // [c][N] plaintexts of the plain operations
  for (size_t i = 0; i < c; i++) {
    table->src[i] = batch[i]->in0;
    if (nin == 2 || plain_op) table->src[c + i] = batch[i]->in1;
    table->dst[i] = batch[i]->out;
    table->flags[i] = 1u;
  }
  if (launch_gather_items(table->src, in_stage, in_words, c * nin, s) != hipSuccess) return all(kHipError);
  if (plain_op && launch_gather_items(table->src + c, pl_stage, pl_words, c, s) != hipSuccess) return all(kHipError);
  // the batch's keys: one for all (the common case), or the distinct ones with each item's index (a per-item selection)
  KeySel ksel(r0.key);
  std::vector<const u64*> keys;
  std::vector<u32> kidx;
  if (r0.kind == 1 || r0.kind == 2) {
    bool one = true;
    for (const CombReq* r : batch) one = one && r->key == r0.key;
    if (!one) {
      kidx.resize(c);
      for (size_t i = 0; i < c; i++) {
        size_t k = 0;
        while (k < keys.size() && keys[k] != batch[i]->key) k++;
        if (k == keys.size()) keys.push_back(batch[i]->key);
        kidx[i] = (u32)k;
      }
      ksel = KeySel();
      ksel.keys = keys.data();
      ksel.nkeys = (u32)keys.size();
      ksel.index = kidx.data();
      ksel.period = c;
    }
  }
  int st = r0.kind == 0 ? ev.multiply(in_stage, 2, in_stage + (squares ? 0 : c * in_words), 2, out_stage, c, s)
         : r0.kind == 1 ? ev.relinearize(in_stage, ksel, out_stage, c, s)
         : r0.kind == 2 ? ev.apply_galois(in_stage, r0.elt, ksel, out_stage, c, s)
         : r0.kind == 5 ? ev.add_plain(in_stage, 2, pl_stage, pl_words, out_stage, c, s)
         : r0.kind == 6 ? ev.sub_plain(in_stage, 2, pl_stage, pl_words, out_stage, c, s)
                        : ev.multiply_plain(in_stage, 2, pl_stage, pl_words, out_stage, c, s);  // monomials included: same bits as the shortcut
  if (st) return all(st);
  if (launch_scatter_items(out_stage, table->dst, out_words, c, s) != hipSuccess) return all(kHipError);
  if (g_throw_transparent && launch_transparent_flags(out_stage, out_words, poly, table->flags, c, s) != hipSuccess) return all(kHipError);
  if (hipStreamSynchronize(s) != hipSuccess) return all(kHipError);
  for (size_t i = 0; i < c; i++) {
    batch[i]->status = kOk;
    batch[i]->nonzero = table->flags[i] ? 1 : 0;
  }
}

// One handle-level operation, possibly executed as part of a batch led by another thread.  On return with status 0 the
// result is in req.out; req.nonzero >= 0 means it is complete on the device and already checked.
int combine_run(EvalObj* e, CombReq& req, hipStream_t s) {
  static const bool enabled = [] {
    const char* env = std::getenv("HIPBFV_NO_COMBINE");
    return !(env && env[0] == '1');
  }();
  if (!enabled) {
    std::vector<CombReq*> one{&req};
    combine_execute(e, one, s);
    return req.status;
  }
  e->comb.run(
      req, kCombineLeaders, kCombineMax,
      // r06: requests of one kind (and Galois element) combine ACROSS keys -- many clients' calls on one evaluator, each with its own
      // keys (the reference passes them per call, run.rs:100-105), run as one per-key batch (Evaluator KeySel)
      [](const CombReq& head, const CombReq& r) { return r.kind == head.kind && r.elt == head.elt; },
      [&](const std::vector<CombReq*>& batch) {
        try {
          combine_execute(e, batch, s);
        } catch (...) {
          (void)hipStreamSynchronize(s);  // nothing of the batch may still be in flight when its owners see the status
          for (CombReq* r : batch) r->status = kOutOfMemory;
        }
      });
  return req.status;
}

// known: the transparent verdict when a combined batch already produced it (and synchronised): 1 / 0; -1 = check here
long finish_result(EvalObj* e, CipherObj* dst, u32 size, u64* buf, size_t words, hipStream_t s, bool check_transparent, int known = -1) {
  u32 flag = 1;
  if (known >= 0) {
    flag = (u32)known;
  } else if (check_transparent && g_throw_transparent) {
    // one word of pinned, device-addressable host memory per host thread: the check kernel writes its verdict there
    thread_local u32* host_flag = nullptr;
    if (!host_flag && hipHostMalloc((void**)&host_flag, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
      host_flag = nullptr;
      (void)hipGetLastError();
      g_buffers.put(buf, words);
      return from_status(kOutOfMemory);
    }
    *host_flag = 1u;
    const size_t poly = (size_t)e->ctx->K() * e->ctx->n();
    long hr = launch_transparent_flag(buf, (size_t)size * poly, poly, host_flag, s) == hipSuccess ? sync_stream(s) : from_status(kHipError);
    if (hr != HIPBFV_S_OK) {
      g_buffers.put(buf, words);
      return hr;
    }
    flag = *(volatile u32*)host_flag;
  } else {
    long hr = sync_stream(s);
    if (hr != HIPBFV_S_OK) {
      g_buffers.put(buf, words);
      return hr;
    }
  }
  if (!flag) {
    g_buffers.put(buf, words);
    return from_status(kTransparent);
  }
  dst->adopt(e->ctx, size, buf, words);
  return HIPBFV_S_OK;
}

}  // namespace

// The C ABI is the library's only interface: with `make HIDDEN=1` (-fvisibility=hidden) everything else stays inside the DSO
// (HISTORY.md section 11: the internal hipbfv:: C++ symbols are otherwise exported and can be interposed by user code).
//
// Exception barrier: no C++ exception may unwind through an extern "C" frame into Rust / ctypes (abort or UB there).  Every
// exported function is a function-try-block that maps what is thrown to the HRESULT SEAL's C layer would return
// (std::bad_alloc / length_error -> E_OUTOFMEMORY, invalid_argument / out_of_range -> E_INVALIDARG, logic_error ->
// COR_E_INVALIDOPERATION, anything else -> E_UNEXPECTED) and records the message for hipbfv_last_error.
#define HIPBFV_BEGIN try { enter_thread();
#define HIPBFV_END                                                                                               \
  }                                                                                                              \
  catch (const std::bad_alloc&) { return fail(HIPBFV_E_OUTOFMEMORY, "out of host memory"); }                      \
  catch (const std::length_error& x) { return fail(HIPBFV_E_OUTOFMEMORY, x.what()); }                             \
  catch (const std::invalid_argument& x) { return fail(HIPBFV_E_INVALIDARG, x.what()); }                          \
  catch (const std::out_of_range& x) { return fail(HIPBFV_E_INVALIDARG, x.what()); }                              \
  catch (const std::logic_error& x) { return fail(HIPBFV_COR_E_INVALIDOPERATION, x.what()); }                     \
  catch (const std::exception& x) { return fail(HIPBFV_E_UNEXPECTED, x.what()); }                                 \
  catch (...) { return fail(HIPBFV_E_UNEXPECTED, "unknown C++ exception"); }

#pragma GCC visibility push(default)
extern "C" {

// ------------------------------------------------------------------ library
long hipbfv_version(uint32_t* major, uint32_t* minor) HIPBFV_BEGIN
  if (!major || !minor) return HIPBFV_E_POINTER;
  *major = 0;
  *minor = 1;
  return HIPBFV_S_OK;
HIPBFV_END

// The macro definitions this library was compiled with (the Makefile and tools/build_variant.sh hand every translation unit
// their own flag string): bench.py folds it into the kernel-source hash that guards the committed PMC figures, so numbers taken
// on a default build are never printed beside a -D variant's timings.
#ifndef HIPBFV_BUILD_FLAGS
#define HIPBFV_BUILD_FLAGS "unknown"
#endif
long hipbfv_build_flags(char* buffer, uint64_t capacity) HIPBFV_BEGIN
  if (!buffer || !capacity) return HIPBFV_E_POINTER;
  const char* flags = HIPBFV_BUILD_FLAGS;
  const size_t c = std::min<size_t>(std::strlen(flags), capacity - 1);
  std::memcpy(buffer, flags, c);
  buffer[c] = 0;
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_last_error(char* buffer, uint64_t capacity) HIPBFV_BEGIN
  if (!buffer || capacity == 0) return HIPBFV_E_POINTER;
  std::strncpy(buffer, tls_error.c_str(), capacity - 1);
  buffer[capacity - 1] = 0;
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_set_device(int device) HIPBFV_BEGIN
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return fail(HIPBFV_E_INVALIDARG, "no such HIP device");
  if (device != g_device.load() && g_live_contexts.load() > 0)
    return fail(HIPBFV_COR_E_INVALIDOPERATION, "hipbfv_set_device: contexts exist on the current device (one device per process; destroy them first)");
  if (device != g_device.load()) g_buffers.drain();  // cached blocks belong to the device they were allocated on
  g_device = device;
  enter_thread();
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_set_throw_on_transparent(bool enabled) HIPBFV_BEGIN
  g_throw_transparent = enabled;
  return HIPBFV_S_OK;
HIPBFV_END

// ------------------------------------------------------------------ Modulus
long Modulus_Create1(uint64_t value, void** out) HIPBFV_BEGIN
  if (!out) return HIPBFV_E_POINTER;
  if (value == 1 || (value >> 61)) return fail(HIPBFV_E_INVALIDARG, "modulus must be 0 or in [2, 2^61)");
  *out = new ModulusObj(value);
  return HIPBFV_S_OK;
HIPBFV_END

long Modulus_Create2(void* copy, void** out) HIPBFV_BEGIN
  ModulusObj* m = as<ModulusObj>(copy, kMagicModulus);
  if (!m || !out) return HIPBFV_E_POINTER;
  *out = new ModulusObj(m->value);
  return HIPBFV_S_OK;
HIPBFV_END

long Modulus_Destroy(void* p) HIPBFV_BEGIN
  ModulusObj* m = as<ModulusObj>(p, kMagicModulus);
  if (!m) return HIPBFV_E_POINTER;
  delete m;
  return HIPBFV_S_OK;
HIPBFV_END

long Modulus_Value(void* p, uint64_t* result) HIPBFV_BEGIN
  ModulusObj* m = as<ModulusObj>(p, kMagicModulus);
  if (!m || !result) return HIPBFV_E_POINTER;
  *result = m->value;
  return HIPBFV_S_OK;
HIPBFV_END

// ------------------------------------------------------------------ CoeffModulus
long CoeffModulus_MaxBitCount(uint64_t n, int sec, int* bit_count) HIPBFV_BEGIN
  if (!bit_count) return HIPBFV_E_POINTER;
  *bit_count = max_coeff_bit_count(n, sec);
  return *bit_count ? HIPBFV_S_OK : fail(HIPBFV_E_INVALIDARG, "unsupported degree / security level");
HIPBFV_END

long CoeffModulus_BFVDefault(uint64_t n, int sec, uint64_t* length, void** coeffs) HIPBFV_BEGIN
  if (!length) return HIPBFV_E_POINTER;
  std::vector<u64> p = default_coeff_modulus(n, sec);
  if (p.empty()) return fail(HIPBFV_E_INVALIDARG, "no default coefficient modulus for these parameters");
  *length = p.size();
  if (!coeffs) return HIPBFV_S_OK;  // SEAL convention: first call queries the length
  for (size_t i = 0; i < p.size(); i++) coeffs[i] = new ModulusObj(p[i]);
  return HIPBFV_S_OK;
HIPBFV_END

long CoeffModulus_Create1(uint64_t n, uint64_t length, int* bit_sizes, void** coeffs) HIPBFV_BEGIN
  if (!bit_sizes || !coeffs) return HIPBFV_E_POINTER;
  if (n < 2 || (n & (n - 1)) || length == 0 || length > 64) return fail(HIPBFV_E_INVALIDARG, "invalid degree or length");
  // per distinct bit size: take the needed count from the descending prime list, hand out smallest first
  std::vector<u64> result(length, 0);
  std::vector<bool> done(length, false);
  for (uint64_t i = 0; i < length; i++) {
    if (done[i]) continue;
    if (bit_sizes[i] < 2 || bit_sizes[i] > 60) return fail(HIPBFV_E_INVALIDARG, "bit sizes must be in [2, 60]");
    size_t need = 0;
    for (uint64_t j = i; j < length; j++) need += bit_sizes[j] == bit_sizes[i];
    std::vector<u64> primes = find_primes(2 * n, bit_sizes[i], need);
    if (primes.size() != need) return fail(HIPBFV_E_INVALIDARG, "failed to find enough qualifying primes");
    for (uint64_t j = i; j < length; j++) {
      if (bit_sizes[j] == bit_sizes[i]) {
        result[j] = primes.back();
        primes.pop_back();
        done[j] = true;
      }
    }
  }
  for (uint64_t i = 0; i < length; i++) coeffs[i] = new ModulusObj(result[i]);
  return HIPBFV_S_OK;
HIPBFV_END

// ------------------------------------------------------------------ EncryptionParameters
long EncParams_Create1(uint8_t scheme, void** out) HIPBFV_BEGIN
  if (!out) return HIPBFV_E_POINTER;
  if (scheme != 1) return fail(HIPBFV_E_INVALIDARG, "only the BFV scheme (1) is supported");
  ParamsObj* p = new ParamsObj();
  p->scheme = scheme;
  *out = p;
  return HIPBFV_S_OK;
HIPBFV_END

long EncParams_Destroy(void* h) HIPBFV_BEGIN
  ParamsObj* p = as<ParamsObj>(h, kMagicParams);
  if (!p) return HIPBFV_E_POINTER;
  delete p;
  return HIPBFV_S_OK;
HIPBFV_END

long EncParams_SetPolyModulusDegree(void* h, uint64_t degree) HIPBFV_BEGIN
  ParamsObj* p = as<ParamsObj>(h, kMagicParams);
  if (!p) return HIPBFV_E_POINTER;
  p->n = degree;
  return HIPBFV_S_OK;
HIPBFV_END

long EncParams_GetPolyModulusDegree(void* h, uint64_t* degree) HIPBFV_BEGIN
  ParamsObj* p = as<ParamsObj>(h, kMagicParams);
  if (!p || !degree) return HIPBFV_E_POINTER;
  *degree = p->n;
  return HIPBFV_S_OK;
HIPBFV_END

long EncParams_SetCoeffModulus(void* h, uint64_t length, void** coeffs) HIPBFV_BEGIN
  ParamsObj* p = as<ParamsObj>(h, kMagicParams);
  if (!p || !coeffs) return HIPBFV_E_POINTER;
  std::vector<u64> v;
  for (uint64_t i = 0; i < length; i++) {
    ModulusObj* m = as<ModulusObj>(coeffs[i], kMagicModulus);
    if (!m) return HIPBFV_E_POINTER;
    v.push_back(m->value);
  }
  p->coeff = v;
  return HIPBFV_S_OK;
HIPBFV_END

long EncParams_GetCoeffModulus(void* h, uint64_t* length, void** coeffs) HIPBFV_BEGIN
  ParamsObj* p = as<ParamsObj>(h, kMagicParams);
  if (!p || !length) return HIPBFV_E_POINTER;
  *length = p->coeff.size();
  if (!coeffs) return HIPBFV_S_OK;
  for (size_t i = 0; i < p->coeff.size(); i++) coeffs[i] = new ModulusObj(p->coeff[i]);
  return HIPBFV_S_OK;
HIPBFV_END

long EncParams_SetPlainModulus1(void* h, void* modulus) HIPBFV_BEGIN
  ParamsObj* p = as<ParamsObj>(h, kMagicParams);
  ModulusObj* m = as<ModulusObj>(modulus, kMagicModulus);
  if (!p || !m) return HIPBFV_E_POINTER;
  p->plain = m->value;
  return HIPBFV_S_OK;
HIPBFV_END

long EncParams_SetPlainModulus2(void* h, uint64_t plain) HIPBFV_BEGIN
  ParamsObj* p = as<ParamsObj>(h, kMagicParams);
  if (!p) return HIPBFV_E_POINTER;
  p->plain = plain;
  return HIPBFV_S_OK;
HIPBFV_END

long EncParams_GetPlainModulus(void* h, void** modulus) HIPBFV_BEGIN
  ParamsObj* p = as<ParamsObj>(h, kMagicParams);
  if (!p || !modulus) return HIPBFV_E_POINTER;
  *modulus = new ModulusObj(p->plain);
  return HIPBFV_S_OK;
HIPBFV_END

long EncParams_GetScheme(void* h, uint8_t* scheme) HIPBFV_BEGIN
  ParamsObj* p = as<ParamsObj>(h, kMagicParams);
  if (!p || !scheme) return HIPBFV_E_POINTER;
  *scheme = p->scheme;
  return HIPBFV_S_OK;
HIPBFV_END

// ------------------------------------------------------------------ SEALContext
static long make_context(u64 n, const std::vector<u64>& coeff, u64 plain, int sec, void** out) {
  if (sec != 0) {
    const int maxbits = max_coeff_bit_count(n, sec);
    int total = 0;
    for (u64 q : coeff) total += 64 - __builtin_clzll(q | 1);
    if (!maxbits || total > maxbits) return fail(HIPBFV_E_INVALIDARG, "parameters do not meet the requested security level");
  }
  std::string err;
  Context* c = Context::create((u32)n, coeff, plain, g_device, &err);
  if (!c) return fail(HIPBFV_E_INVALIDARG, err.c_str());
  ContextObj* o = new ContextObj();
  g_live_contexts.fetch_add(1);
  o->ctx.reset(c, [](Context* x) {
    delete x;
    g_live_contexts.fetch_sub(1);
  });
  *out = o;
  return HIPBFV_S_OK;
}

long SEALContext_Create(void* params, bool expand_mod_chain, int sec_level, void** context) HIPBFV_BEGIN
  ParamsObj* p = as<ParamsObj>(params, kMagicParams);
  if (!p || !context) return HIPBFV_E_POINTER;
  if (p->n > 0xFFFFFFFFull) return fail(HIPBFV_E_INVALIDARG, "invalid degree");
  long hr = make_context(p->n, p->coeff, p->plain, sec_level, context);
  // lower levels are created lazily on the first Evaluator_ModSwitchToNext; without the chain that call fails
  if (hr == HIPBFV_S_OK) static_cast<ContextObj*>(*context)->ctx->set_chain_enabled(expand_mod_chain);
  return hr;
HIPBFV_END

long SEALContext_Destroy(void* h) HIPBFV_BEGIN
  ContextObj* c = as<ContextObj>(h, kMagicContext);
  if (!c) return HIPBFV_E_POINTER;
  delete c;
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_Context_Create(uint64_t n, const uint64_t* coeff, uint64_t count, uint64_t plain, void** context) HIPBFV_BEGIN
  if (!coeff || !context) return HIPBFV_E_POINTER;
  if (n > 0xFFFFFFFFull) return fail(HIPBFV_E_INVALIDARG, "invalid degree");
  return make_context(n, std::vector<u64>(coeff, coeff + count), plain, 0, context);
HIPBFV_END

long hipbfv_Context_Info(void* h, uint64_t* n, uint64_t* K, uint64_t* KK, uint64_t* t) HIPBFV_BEGIN
  ContextObj* c = as<ContextObj>(h, kMagicContext);
  if (!c) return HIPBFV_E_POINTER;
  if (n) *n = c->ctx->n();
  if (K) *K = c->ctx->K();
  if (KK) *KK = c->ctx->KK();
  if (t) *t = c->ctx->t();
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_Context_GetPrime(void* h, uint64_t index, uint64_t* value) HIPBFV_BEGIN
  ContextObj* c = as<ContextObj>(h, kMagicContext);
  if (!c || !value) return HIPBFV_E_POINTER;
  if (index >= c->ctx->key_primes().size()) return fail(HIPBFV_E_INVALIDARG, "prime index out of range");
  *value = c->ctx->key_primes()[index];
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_Context_AuxBase(void* h, uint64_t* count, uint64_t* primes, uint64_t capacity, int* fp64_base) HIPBFV_BEGIN
  ContextObj* c = as<ContextObj>(h, kMagicContext);
  if (!c || !count) return HIPBFV_E_POINTER;
  const hipbfv::DevCtx& d = c->ctx->host();
  *count = d.S;
  if (fp64_base) *fp64_base = (d.aux_f64 ? 1 : 0) | (d.pack_mul ? 2 : 0) | (d.pack_ks ? 4 : 0) | (d.conv_grid ? 8 : 0) | (d.aux_mixed ? 16 : 0) | (d.pack_mul == 2 ? 32 : 0) | (d.pack_ks == 2 ? 64 : 0);
  if (primes) {
    if (capacity < d.S) return fail(HIPBFV_E_INVALIDARG, "capacity too small");
    for (uint32_t j = 0; j < d.S; j++) primes[j] = d.mod[d.KK + j].q;
  }
  return HIPBFV_S_OK;
HIPBFV_END

// ------------------------------------------------------------------ Plaintext
long Plaintext_Create1(void* pool, void** out) HIPBFV_BEGIN
  (void)pool;
  if (!out) return HIPBFV_E_POINTER;
  *out = new PlainObj();
  return HIPBFV_S_OK;
HIPBFV_END

// Plaintext from SEAL's polynomial string "7FFx^3 + 1x^1 + 3" (plaintext_ciphertext.rs:180-217): hexadecimal
// coefficients, decimal exponents, strictly decreasing, terms separated by " + ", constant term without "x^"
long Plaintext_Create4(char* hex_poly, void* pool, void** out) HIPBFV_BEGIN
  (void)pool;
  if (!hex_poly || !out) return HIPBFV_E_POINTER;
  std::vector<std::pair<u64, u64>> terms;  // (exponent, coefficient)
  const char* p = hex_poly;
  auto bad = [] { return fail(HIPBFV_E_INVALIDARG, "unable to parse hex_poly"); };
  while (*p) {
    u64 coeff = 0;
    int digits = 0;
    for (;; p++, digits++) {
      int v;
      if (*p >= '0' && *p <= '9') v = *p - '0';
      else if (*p >= 'a' && *p <= 'f') v = *p - 'a' + 10;
      else if (*p >= 'A' && *p <= 'F') v = *p - 'A' + 10;
      else break;
      if (coeff >> 60) return bad();
      coeff = (coeff << 4) | (u64)v;
    }
    if (!digits) return bad();
    u64 expo = 0;
    if (*p == 'x') {
      if (p[1] != '^') return bad();
      p += 2;
      int ed = 0;
      for (; *p >= '0' && *p <= '9'; p++, ed++) {
        if (expo > (1u << 17)) return bad();  // SEAL_POLY_MOD_DEGREE_MAX = 131072 bounds the allocation below
        expo = expo * 10 + (u64)(*p - '0');
      }
      if (!ed || !expo) return bad();
    }
    if (!terms.empty() && expo >= terms.back().first) return bad();
    terms.emplace_back(expo, coeff);
    if (!*p) break;
    if (expo == 0 || std::strncmp(p, " + ", 3) != 0) return bad();
    p += 3;
    if (!*p) return bad();
  }
  PlainObj* n = new PlainObj();
  if (!terms.empty()) {
    // SEAL sizes the plaintext by its significant coefficients
    size_t top = 0;
    for (auto& t : terms)
      if (t.second) top = std::max<size_t>(top, t.first + 1);
    n->coeffs.assign(top, 0);
    for (auto& t : terms)
      if (t.second) n->coeffs[t.first] = t.second;
  }
  *out = n;
  return HIPBFV_S_OK;
HIPBFV_END

long Plaintext_Create5(void* copy, void** out) HIPBFV_BEGIN
  PlainObj* p = as<PlainObj>(copy, kMagicPlain);
  if (!p || !out) return HIPBFV_E_POINTER;
  PlainObj* n = new PlainObj();
  n->coeffs = p->coeffs;
  *out = n;
  return HIPBFV_S_OK;
HIPBFV_END

long Plaintext_Destroy(void* h) HIPBFV_BEGIN
  PlainObj* p = as<PlainObj>(h, kMagicPlain);
  if (!p) return HIPBFV_E_POINTER;
  delete p;
  return HIPBFV_S_OK;
HIPBFV_END

long Plaintext_CoeffCount(void* h, uint64_t* count) HIPBFV_BEGIN
  PlainObj* p = as<PlainObj>(h, kMagicPlain);
  if (!p || !count) return HIPBFV_E_POINTER;
  *count = p->coeffs.size();
  return HIPBFV_S_OK;
HIPBFV_END

long Plaintext_CoeffAt(void* h, uint64_t index, uint64_t* coeff) HIPBFV_BEGIN
  PlainObj* p = as<PlainObj>(h, kMagicPlain);
  if (!p || !coeff) return HIPBFV_E_POINTER;
  if (index >= p->coeffs.size()) return fail(HIPBFV_E_INVALIDARG, "coefficient index out of range");
  *coeff = p->coeffs[index];
  return HIPBFV_S_OK;
HIPBFV_END

long Plaintext_SetCoeffAt(void* h, uint64_t index, uint64_t value) HIPBFV_BEGIN
  PlainObj* p = as<PlainObj>(h, kMagicPlain);
  if (!p) return HIPBFV_E_POINTER;
  if (index >= p->coeffs.size()) return fail(HIPBFV_E_INVALIDARG, "coefficient index out of range");
  p->coeffs[index] = value;
  return HIPBFV_S_OK;
HIPBFV_END

long Plaintext_Resize(void* h, uint64_t count) HIPBFV_BEGIN
  PlainObj* p = as<PlainObj>(h, kMagicPlain);
  if (!p) return HIPBFV_E_POINTER;
  if (count > (1u << 20)) return fail(HIPBFV_E_INVALIDARG, "plaintext too large");
  p->coeffs.resize(count, 0);
  return HIPBFV_S_OK;
HIPBFV_END

long Plaintext_IsNTTForm(void* h, bool* is_ntt) HIPBFV_BEGIN
  PlainObj* p = as<PlainObj>(h, kMagicPlain);
  if (!p || !is_ntt) return HIPBFV_E_POINTER;
  *is_ntt = false;
  return HIPBFV_S_OK;
HIPBFV_END

// ------------------------------------------------------------------ Ciphertext
long Ciphertext_Create1(void* pool, void** out) HIPBFV_BEGIN
  (void)pool;
  if (!out) return HIPBFV_E_POINTER;
  *out = new CipherObj();
  return HIPBFV_S_OK;
HIPBFV_END

long Ciphertext_Create2(void* copy, void** out) HIPBFV_BEGIN
  CipherObj* c = as<CipherObj>(copy, kMagicCipher);
  if (!c || !out) return HIPBFV_E_POINTER;
  CipherObj* n = new CipherObj();
  if (c->dev) {
    u64* buf = g_buffers.get(c->words);
    if (!buf) {
      delete n;
      return from_status(kOutOfMemory);
    }
    if (copy_d2d(buf, c->dev, c->words * sizeof(u64)) != hipSuccess) {
      g_buffers.put(buf, c->words);
      delete n;
      return from_status(kHipError);
    }
    n->adopt(c->ctx, c->size, buf, c->words);
  }
  *out = n;
  return HIPBFV_S_OK;
HIPBFV_END

long Ciphertext_Destroy(void* h) HIPBFV_BEGIN
  CipherObj* c = as<CipherObj>(h, kMagicCipher);
  if (!c) return HIPBFV_E_POINTER;
  delete c;
  return HIPBFV_S_OK;
HIPBFV_END

long Ciphertext_Size(void* h, uint64_t* size) HIPBFV_BEGIN
  CipherObj* c = as<CipherObj>(h, kMagicCipher);
  if (!c || !size) return HIPBFV_E_POINTER;
  *size = c->size;
  return HIPBFV_S_OK;
HIPBFV_END

long Ciphertext_CoeffModulusSize(void* h, uint64_t* k) HIPBFV_BEGIN
  CipherObj* c = as<CipherObj>(h, kMagicCipher);
  if (!c || !k) return HIPBFV_E_POINTER;
  *k = c->ctx ? c->ctx->K() : 0;
  return HIPBFV_S_OK;
HIPBFV_END

long Ciphertext_PolyModulusDegree(void* h, uint64_t* n) HIPBFV_BEGIN
  CipherObj* c = as<CipherObj>(h, kMagicCipher);
  if (!c || !n) return HIPBFV_E_POINTER;
  *n = c->ctx ? c->ctx->n() : 0;
  return HIPBFV_S_OK;
HIPBFV_END

static long ensure_host(CipherObj* c) {
  // reads of one shared ciphertext from several host threads are legal in SEAL (GetDataAt / Save are const): the lazy
  // mirror is filled under a lock, and host_valid is published after the data
  std::lock_guard<std::mutex> g(c->host_mu);
  if (c->host_valid) return HIPBFV_S_OK;
  c->host.resize(c->words);
  if (c->words && hipMemcpy(c->host.data(), c->dev, c->words * sizeof(u64), hipMemcpyDeviceToHost) != hipSuccess)
    return from_status(kHipError);
  c->host_valid = true;
  return HIPBFV_S_OK;
}

long Ciphertext_GetDataAt1(void* h, uint64_t index, uint64_t* data) HIPBFV_BEGIN
  CipherObj* c = as<CipherObj>(h, kMagicCipher);
  if (!c || !data) return HIPBFV_E_POINTER;
  if (index >= c->words) return fail(HIPBFV_E_INVALIDARG, "index out of range");
  long hr = ensure_host(c);
  if (hr != HIPBFV_S_OK) return hr;
  *data = c->host[index];
  return HIPBFV_S_OK;
HIPBFV_END

long Ciphertext_GetDataAt2(void* h, uint64_t poly_index, uint64_t coeff_index, uint64_t* data) HIPBFV_BEGIN
  CipherObj* c = as<CipherObj>(h, kMagicCipher);
  if (!c || !data) return HIPBFV_E_POINTER;
  if (!c->ctx || poly_index >= c->size || coeff_index >= c->ctx->n()) return fail(HIPBFV_E_INVALIDARG, "index out of range");
  long hr = ensure_host(c);
  if (hr != HIPBFV_S_OK) return hr;
  const size_t K = c->ctx->K(), n = c->ctx->n();
  for (size_t i = 0; i < K; i++) data[i] = c->host[(poly_index * K + i) * n + coeff_index];
  return HIPBFV_S_OK;
HIPBFV_END

long Ciphertext_IsNTTForm(void* h, bool* is_ntt) HIPBFV_BEGIN
  CipherObj* c = as<CipherObj>(h, kMagicCipher);
  if (!c || !is_ntt) return HIPBFV_E_POINTER;
  *is_ntt = false;  // BFV ciphertexts stay in coefficient form between operations (evaluator_base.rs:46-53)
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_Ciphertext_Assign(void* h, void* context, uint64_t size, const uint64_t* host_data) HIPBFV_BEGIN
  CipherObj* c = as<CipherObj>(h, kMagicCipher);
  ContextObj* x = as<ContextObj>(context, kMagicContext);
  if (!c || !x || !host_data) return HIPBFV_E_POINTER;
  if (size < 2 || size > 16) return fail(HIPBFV_E_INVALIDARG, "ciphertext size must be in [2, 16]");
  const size_t words = x->ctx->ct_words(size);
  const size_t K = x->ctx->K(), n = x->ctx->n();
  for (size_t p = 0; p < size * K; p++) {
    const u64 q = x->ctx->key_primes()[p % K];
    for (size_t k = 0; k < n; k++)
      if (host_data[p * n + k] >= q) return fail(HIPBFV_E_INVALIDARG, "ciphertext coefficient is not reduced modulo its prime");
  }
  u64* buf = g_buffers.get(words);
  if (!buf) return from_status(kOutOfMemory);
  if (hipMemcpy(buf, host_data, words * sizeof(u64), hipMemcpyHostToDevice) != hipSuccess) {
    g_buffers.put(buf, words);
    return from_status(kHipError);
  }
  c->adopt(x->ctx, (u32)size, buf, words);
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_Ciphertext_Export(void* h, uint64_t* host_data, uint64_t capacity_words) HIPBFV_BEGIN
  CipherObj* c = as<CipherObj>(h, kMagicCipher);
  if (!c || !host_data) return HIPBFV_E_POINTER;
  if (capacity_words < c->words) return fail(HIPBFV_E_INVALIDARG, "buffer too small");
  if (c->words && hipMemcpy(host_data, c->dev, c->words * sizeof(u64), hipMemcpyDeviceToHost) != hipSuccess)
    return from_status(kHipError);
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_Ciphertext_DevicePtr(void* h, uint64_t** device_ptr) HIPBFV_BEGIN
  CipherObj* c = as<CipherObj>(h, kMagicCipher);
  if (!c || !device_ptr) return HIPBFV_E_POINTER;
  *device_ptr = (uint64_t*)c->dev;
  return HIPBFV_S_OK;
HIPBFV_END

// ------------------------------------------------------------------ KSwitchKeys
long KSwitchKeys_Create1(void** out) HIPBFV_BEGIN
  if (!out) return HIPBFV_E_POINTER;
  *out = new KeysObj();
  return HIPBFV_S_OK;
HIPBFV_END

long KSwitchKeys_Create2(void* copy, void** out) HIPBFV_BEGIN
  KeysObj* k = as<KeysObj>(copy, kMagicKeys);
  if (!k || !out) return HIPBFV_E_POINTER;
  KeysObj* n = new KeysObj();
  n->ctx = k->ctx;
  for (auto& kv : k->keys) {
    u64* buf = g_buffers.get(k->ctx->key_words());
    if (!buf || copy_d2d(buf, kv.second, k->ctx->key_words() * sizeof(u64)) != hipSuccess) {
      if (buf) g_buffers.put(buf, k->ctx->key_words());
      delete n;
      return from_status(buf ? kHipError : kOutOfMemory);
    }
    n->keys[kv.first] = buf;
  }
  *out = n;
  return HIPBFV_S_OK;
HIPBFV_END

long KSwitchKeys_Destroy(void* h) HIPBFV_BEGIN
  KeysObj* k = as<KeysObj>(h, kMagicKeys);
  if (!k) return HIPBFV_E_POINTER;
  delete k;
  return HIPBFV_S_OK;
HIPBFV_END

static long assign_key(KeysObj* k, ContextObj* x, u32 index, const uint64_t* host_data) {
  if (x->ctx->KK() < 2) return fail(HIPBFV_E_INVALIDARG, "these parameters do not support key switching");
  if (k->ctx && k->ctx.get() != x->ctx.get()) return fail(HIPBFV_E_INVALIDARG, "keys belong to a different context");
  const size_t words = x->ctx->key_words();
  const size_t KK = x->ctx->KK(), n = x->ctx->n();
  for (size_t p = 0; p < words / n; p++) {
    const u64 q = x->ctx->key_primes()[p % KK];
    for (size_t i = 0; i < n; i++)
      if (host_data[p * n + i] >= q) return fail(HIPBFV_E_INVALIDARG, "key coefficient is not reduced modulo its prime");
  }
  u64* buf = g_buffers.get(words);
  if (!buf) return from_status(kOutOfMemory);
  if (hipMemcpy(buf, host_data, words * sizeof(u64), hipMemcpyHostToDevice) != hipSuccess) {
    g_buffers.put(buf, words);
    return from_status(kHipError);
  }
  k->ctx = x->ctx;
  auto it = k->keys.find(index);
  if (it != k->keys.end()) g_buffers.put(it->second, words);
  k->keys[index] = buf;
  return HIPBFV_S_OK;
}

long hipbfv_KSwitchKeys_AssignRelin(void* h, void* context, const uint64_t* host_data) HIPBFV_BEGIN
  KeysObj* k = as<KeysObj>(h, kMagicKeys);
  ContextObj* x = as<ContextObj>(context, kMagicContext);
  if (!k || !x || !host_data) return HIPBFV_E_POINTER;
  return assign_key(k, x, 0, host_data);
HIPBFV_END

long hipbfv_KSwitchKeys_AssignGalois(void* h, void* context, uint32_t elt, const uint64_t* host_data) HIPBFV_BEGIN
  KeysObj* k = as<KeysObj>(h, kMagicKeys);
  ContextObj* x = as<ContextObj>(context, kMagicContext);
  if (!k || !x || !host_data) return HIPBFV_E_POINTER;
  if (!(elt & 1) || elt >= 2 * x->ctx->n()) return fail(HIPBFV_E_INVALIDARG, "invalid Galois element");
  return assign_key(k, x, (elt - 1) >> 1, host_data);
HIPBFV_END

long hipbfv_KSwitchKeys_DevicePtr(void* h, uint64_t index, uint64_t** device_ptr) HIPBFV_BEGIN
  KeysObj* k = as<KeysObj>(h, kMagicKeys);
  if (!k || !device_ptr) return HIPBFV_E_POINTER;
  *device_ptr = (uint64_t*)k->find((u32)index);
  return *device_ptr ? HIPBFV_S_OK : fail(HIPBFV_E_INVALIDARG, "key not present");
HIPBFV_END

// ------------------------------------------------------------------ Evaluator (handle level, synchronous)
long Evaluator_Create(void* context, void** out) HIPBFV_BEGIN
  ContextObj* x = as<ContextObj>(context, kMagicContext);
  if (!x || !out) return HIPBFV_E_POINTER;
  EvalObj* e = new EvalObj();
  e->ctx = x->ctx;
  e->ev.reset(new Evaluator(x->ctx.get()));
  *out = e;
  return HIPBFV_S_OK;
HIPBFV_END

long Evaluator_Destroy(void* h) HIPBFV_BEGIN
  EvalObj* e = as<EvalObj>(h, kMagicEval);
  if (!e) return HIPBFV_E_POINTER;
  delete e;
  return HIPBFV_S_OK;
HIPBFV_END

long Evaluator_Negate(void* h, void* a, void* dst) HIPBFV_BEGIN
  EvalObj* e = as<EvalObj>(h, kMagicEval);
  CipherObj *x = as<CipherObj>(a, kMagicCipher), *d = as<CipherObj>(dst, kMagicCipher);
  if (!e || !x || !d) return HIPBFV_E_POINTER;
  if (EvalObj* le = level_eval(e, x->ctx)) e = le;
  if (!same_context(x, e)) return fail(HIPBFV_E_INVALIDARG, "encrypted is not valid for encryption parameters");
  hipStream_t s = thread_stream();
  u64* buf = g_buffers.get(x->words);
  if (!buf) return from_status(kOutOfMemory);
  int st = e->ev->negate(x->dev, buf, x->size, 1, s);
  if (st) {
    g_buffers.put(buf, x->words);
    return from_status(st);
  }
  return finish_result(e, d, x->size, buf, x->words, s, false);
HIPBFV_END

// add/sub with SEAL's size rule: the result has max(size) polynomials; extra ones are copied (negated for sub)
static long add_sub(void* h, void* a, void* b, void* dst, bool sub) {
  EvalObj* e = as<EvalObj>(h, kMagicEval);
  CipherObj *x = as<CipherObj>(a, kMagicCipher), *y = as<CipherObj>(b, kMagicCipher), *d = as<CipherObj>(dst, kMagicCipher);
  if (!e || !x || !y || !d) return HIPBFV_E_POINTER;
  if (EvalObj* le = level_eval(e, x->ctx)) e = le;
  if (!same_context(x, e) || !same_context(y, e)) return fail(HIPBFV_E_INVALIDARG, "encrypted is not valid for encryption parameters");
  hipStream_t s = thread_stream();
  const u32 smax = std::max(x->size, y->size), smin = std::min(x->size, y->size);
  const size_t words = e->ctx->ct_words(smax), common = e->ctx->ct_words(smin);
  u64* buf = g_buffers.get(words);
  if (!buf) return from_status(kOutOfMemory);
  if (smax == 2 && smin == 2) {  // the common case joins concurrent callers' batches
    CombReq rq;
    rq.kind = sub ? 4 : 3, rq.in0 = x->dev, rq.in1 = y->dev, rq.out = buf;
    const int st2 = combine_run(e, rq, s);
    if (st2) {
      g_buffers.put(buf, words);
      return from_status(st2);
    }
    return finish_result(e, d, 2, buf, words, s, true, rq.nonzero);
  }
  int st = sub ? e->ev->sub(x->dev, y->dev, buf, smin, 1, s) : e->ev->add(x->dev, y->dev, buf, smin, 1, s);
  if (st == kOk && smax > smin) {
    const CipherObj* big = x->size > y->size ? x : y;
    if (sub && big == y)  // negate the extra polynomials of the subtrahend
      st = e->ev->negate(y->dev + common, buf + common, smax - smin, 1, s);
    else if (hipMemcpyAsync(buf + common, big->dev + common, (words - common) * sizeof(u64), hipMemcpyDeviceToDevice, s) != hipSuccess)
      st = kHipError;
  }
  if (st) {
    g_buffers.put(buf, words);
    return from_status(st);
  }
  return finish_result(e, d, smax, buf, words, s, true);
}

long Evaluator_Add(void* h, void* a, void* b, void* dst) HIPBFV_BEGIN return add_sub(h, a, b, dst, false); HIPBFV_END
long Evaluator_Sub(void* h, void* a, void* b, void* dst) HIPBFV_BEGIN return add_sub(h, a, b, dst, true); HIPBFV_END

long Evaluator_AddMany(void* h, uint64_t count, void** cts, void* dst) HIPBFV_BEGIN
  if (!cts) return HIPBFV_E_POINTER;
  if (count == 0) return fail(HIPBFV_E_INVALIDARG, "encrypteds cannot be empty");
  void* acc = nullptr;
  long hr = Ciphertext_Create2(cts[0], &acc);
  if (hr != HIPBFV_S_OK) return hr;
  for (uint64_t i = 1; i < count && hr == HIPBFV_S_OK; i++) hr = Evaluator_Add(h, acc, cts[i], acc);
  if (hr == HIPBFV_S_OK) {
    CipherObj *a = as<CipherObj>(acc, kMagicCipher), *d = as<CipherObj>(dst, kMagicCipher);
    if (!d)
      hr = HIPBFV_E_POINTER;
    else {
      u64* buf = a->dev;
      const size_t w = a->words;
      a->dev = nullptr;
      a->words = 0;
      d->adopt(a->ctx, a->size, buf, w);
    }
  }
  Ciphertext_Destroy(acc);
  return hr;
HIPBFV_END

long Evaluator_Multiply(void* h, void* a, void* b, void* dst, void* pool) HIPBFV_BEGIN
  (void)pool;
  EvalObj* e = as<EvalObj>(h, kMagicEval);
  CipherObj *x = as<CipherObj>(a, kMagicCipher), *y = as<CipherObj>(b, kMagicCipher), *d = as<CipherObj>(dst, kMagicCipher);
  if (!e || !x || !y || !d) return HIPBFV_E_POINTER;
  if (EvalObj* le = level_eval(e, x->ctx)) e = le;
  if (!same_context(x, e) || !same_context(y, e)) return fail(HIPBFV_E_INVALIDARG, "encrypted is not valid for encryption parameters");
  hipStream_t s = thread_stream();
  const u32 sd = x->size + y->size - 1;
  const size_t words = e->ctx->ct_words(sd);
  u64* buf = g_buffers.get(words);
  if (!buf) return from_status(kOutOfMemory);
  int st, known = -1;
  if (x->size == 2 && y->size == 2) {
    CombReq rq;
    rq.kind = 0, rq.in0 = x->dev, rq.in1 = y->dev, rq.out = buf;
    st = combine_run(e, rq, s);
    known = rq.nonzero;
  } else {
    st = e->ev->multiply(x->dev, x->size, y->dev, y->size, buf, 1, s);
  }
  if (st) {
    g_buffers.put(buf, words);
    return from_status(st);
  }
  return finish_result(e, d, sd, buf, words, s, true, known);
HIPBFV_END

long Evaluator_Square(void* h, void* a, void* dst, void* pool) HIPBFV_BEGIN return Evaluator_Multiply(h, a, a, dst, pool); HIPBFV_END

long Evaluator_Relinearize(void* h, void* a, void* keys, void* dst, void* pool) HIPBFV_BEGIN
  (void)pool;
  EvalObj* e = as<EvalObj>(h, kMagicEval);
  CipherObj *x = as<CipherObj>(a, kMagicCipher), *d = as<CipherObj>(dst, kMagicCipher);
  KeysObj* k = as<KeysObj>(keys, kMagicKeys);
  if (!e || !x || !d || !k) return HIPBFV_E_POINTER;
  if (EvalObj* le = level_eval(e, x->ctx)) e = le;
  if (!same_context(x, e)) return fail(HIPBFV_E_INVALIDARG, "encrypted is not valid for encryption parameters");
  if (x->size == 2) {  // nothing to do: SEAL returns a copy
    if (d == x) return HIPBFV_S_OK;
    u64* buf = g_buffers.get(x->words);
    if (!buf) return from_status(kOutOfMemory);
    if (copy_d2d(buf, x->dev, x->words * sizeof(u64)) != hipSuccess) {
      g_buffers.put(buf, x->words);
      return from_status(kHipError);
    }
    d->adopt(x->ctx, 2, buf, x->words);
    return HIPBFV_S_OK;
  }
  // seal_fhe creates exactly one relinearization key (key_generator.rs:450-452): only size 3 -> 2
  if (x->size != 3) return fail(HIPBFV_E_INVALIDARG, "not enough relinearization keys");
  const u64* rkey = level_key(k, e, 0);
  if (!rkey) return fail(HIPBFV_E_INVALIDARG, "relin_keys is not valid for encryption parameters");
  hipStream_t s = thread_stream();
  const size_t words = e->ctx->ct_words(2);
  u64* buf = g_buffers.get(words);
  if (!buf) return from_status(kOutOfMemory);
  CombReq rq;
  rq.kind = 1, rq.in0 = x->dev, rq.key = rkey, rq.out = buf;
  int st = combine_run(e, rq, s);
  if (st) {
    g_buffers.put(buf, words);
    return from_status(st);
  }
  return finish_result(e, d, 2, buf, words, s, true, rq.nonzero);
HIPBFV_END

// SEAL Evaluator::multiply_many: pairwise products with relinearisation, results appended to the work list
long Evaluator_MultiplyMany(void* h, uint64_t count, void** cts, void* keys, void* dst, void* pool) HIPBFV_BEGIN
  if (!cts) return HIPBFV_E_POINTER;
  if (count == 0) return fail(HIPBFV_E_INVALIDARG, "encrypteds vector must not be empty");
  CipherObj* d = as<CipherObj>(dst, kMagicCipher);
  if (!d) return HIPBFV_E_POINTER;
  std::vector<void*> work;
  long hr = HIPBFV_S_OK;
  auto cleanup = [&]() {
    for (void* w : work) Ciphertext_Destroy(w);
  };
  if (count == 1) {
    void* c = nullptr;
    hr = Ciphertext_Create2(cts[0], &c);
    if (hr != HIPBFV_S_OK) return hr;
    work.push_back(c);
  } else {
    for (uint64_t i = 0; i + 1 < count && hr == HIPBFV_S_OK; i += 2) {
      void* t = nullptr;
      hr = Ciphertext_Create1(nullptr, &t);
      if (hr != HIPBFV_S_OK) break;
      work.push_back(t);
      hr = Evaluator_Multiply(h, cts[i], cts[i + 1], t, pool);
      if (hr == HIPBFV_S_OK) hr = Evaluator_Relinearize(h, t, keys, t, pool);
    }
    if (hr == HIPBFV_S_OK && (count & 1)) {
      void* c = nullptr;
      hr = Ciphertext_Create2(cts[count - 1], &c);
      if (hr == HIPBFV_S_OK) work.push_back(c);
    }
    for (size_t i = 0; hr == HIPBFV_S_OK && i + 1 < work.size(); i += 2) {
      void* t = nullptr;
      hr = Ciphertext_Create1(nullptr, &t);
      if (hr != HIPBFV_S_OK) break;
      work.push_back(t);
      hr = Evaluator_Multiply(h, work[i], work[i + 1], t, pool);
      if (hr == HIPBFV_S_OK) hr = Evaluator_Relinearize(h, t, keys, t, pool);
    }
  }
  if (hr == HIPBFV_S_OK) {
    CipherObj* last = as<CipherObj>(work.back(), kMagicCipher);
    u64* buf = last->dev;
    const size_t w = last->words;
    last->dev = nullptr;
    last->words = 0;
    d->adopt(last->ctx, last->size, buf, w);
  }
  cleanup();
  return hr;
HIPBFV_END

long Evaluator_Exponentiate(void* h, void* a, uint64_t exponent, void* keys, void* dst, void* pool) HIPBFV_BEGIN
  if (exponent == 0) return fail(HIPBFV_E_INVALIDARG, "exponent cannot be 0");
  if (exponent > 4096) return fail(HIPBFV_E_INVALIDARG, "exponent too large");
  std::vector<void*> v((size_t)exponent, a);
  return Evaluator_MultiplyMany(h, exponent, v.data(), keys, dst, pool);
HIPBFV_END

static long plain_to_device(EvalObj* e, PlainObj* p, u64** out, size_t* nonzero, size_t* last_nonzero) {
  const size_t n = e->ctx->n();
  if (p->coeffs.size() > n) return fail(HIPBFV_E_INVALIDARG, "plain is not valid for encryption parameters");
  // staged through pinned memory of the calling thread and copied on its own stream: a plain hipMemcpy from pageable memory
  // goes through the null stream, where the copies of concurrent callers queue behind one another
  thread_local u64* padded = nullptr;
  thread_local size_t padded_words = 0;
  if (padded_words < n) {
    if (padded) (void)hipHostFree(padded);
    padded = nullptr;
    padded_words = 0;
    if (hipHostMalloc((void**)&padded, n * sizeof(u64), hipHostMallocPortable) != hipSuccess) {
      padded = nullptr;
      (void)hipGetLastError();
      return from_status(kOutOfMemory);
    }
    padded_words = n;
  }
  *nonzero = 0;
  *last_nonzero = 0;
  for (size_t i = 0; i < p->coeffs.size(); i++) {
    if (p->coeffs[i] >= e->ctx->t()) return fail(HIPBFV_E_INVALIDARG, "plain is not valid for encryption parameters");
    padded[i] = p->coeffs[i];
    if (padded[i]) {
      (*nonzero)++;
      *last_nonzero = i;
    }
  }
  for (size_t i = p->coeffs.size(); i < n; i++) padded[i] = 0;
  u64* buf = g_buffers.get(n);
  if (!buf) return from_status(kOutOfMemory);
  hipStream_t ps = thread_stream();
  if (hipMemcpyAsync(buf, padded, n * sizeof(u64), hipMemcpyHostToDevice, ps) != hipSuccess || hipStreamSynchronize(ps) != hipSuccess) {
    g_buffers.put(buf, n);
    return from_status(kHipError);
  }
  *out = buf;
  return HIPBFV_S_OK;
}

static long plain_op(void* h, void* a, void* plain, void* dst, int which) {
  EvalObj* e = as<EvalObj>(h, kMagicEval);
  CipherObj *x = as<CipherObj>(a, kMagicCipher), *d = as<CipherObj>(dst, kMagicCipher);
  PlainObj* p = as<PlainObj>(plain, kMagicPlain);
  if (!e || !x || !d || !p) return HIPBFV_E_POINTER;
  if (EvalObj* le = level_eval(e, x->ctx)) e = le;
  if (!same_context(x, e)) return fail(HIPBFV_E_INVALIDARG, "encrypted is not valid for encryption parameters");
  u64* dplain = nullptr;
  size_t nonzero = 0, last = 0;
  long hr = plain_to_device(e, p, &dplain, &nonzero, &last);
  if (hr != HIPBFV_S_OK) return hr;
  hipStream_t s = thread_stream();
  u64* buf = g_buffers.get(x->words);
  if (!buf) {
    g_buffers.put(dplain, e->ctx->n());
    return from_status(kOutOfMemory);
  }
  if (x->size == 2) {  // the common case joins concurrent callers' batches
    CombReq rq;
    rq.kind = 5 + which, rq.in0 = x->dev, rq.in1 = dplain, rq.out = buf;
    if (which == 2 && nonzero == 1) rq.mono_coeff = p->coeffs[last], rq.mono_exp = (u32)last;
    const int st2 = combine_run(e, rq, s);
    if (st2) {
      g_buffers.put(buf, x->words);
      g_buffers.put(dplain, e->ctx->n());
      return from_status(st2);
    }
    hr = finish_result(e, d, 2, buf, x->words, s, true, rq.nonzero);
    g_buffers.put(dplain, e->ctx->n());
    return hr;
  }
  int st;
  if (which == 0)
    st = e->ev->add_plain(x->dev, x->size, dplain, 0, buf, 1, s);
  else if (which == 1)
    st = e->ev->sub_plain(x->dev, x->size, dplain, 0, buf, 1, s);
  else if (nonzero == 1)
    st = e->ev->multiply_plain_mono(x->dev, x->size, p->coeffs[last], (u32)last, buf, 1, s);
  else
    st = e->ev->multiply_plain(x->dev, x->size, dplain, 0, buf, 1, s);
  if (st) {
    (void)hipStreamSynchronize(s);
    g_buffers.put(buf, x->words);
    g_buffers.put(dplain, e->ctx->n());
    return from_status(st);
  }
  hr = finish_result(e, d, x->size, buf, x->words, s, true);
  g_buffers.put(dplain, e->ctx->n());
  return hr;
}

long Evaluator_AddPlain(void* h, void* a, void* plain, void* dst) HIPBFV_BEGIN return plain_op(h, a, plain, dst, 0); HIPBFV_END
long Evaluator_SubPlain(void* h, void* a, void* plain, void* dst) HIPBFV_BEGIN return plain_op(h, a, plain, dst, 1); HIPBFV_END
long Evaluator_MultiplyPlain(void* h, void* a, void* plain, void* dst, void* pool) HIPBFV_BEGIN
  (void)pool;
  return plain_op(h, a, plain, dst, 2);
HIPBFV_END

// one Galois automorphism + key switch on a handle (SEAL apply_galois_inplace)
static long galois_handle(EvalObj* e, CipherObj* x, u32 elt, KeysObj* k, CipherObj* d) {
  const u64* key = level_key(k, e, (elt - 1) >> 1);
  if (!key) return from_status(kNoKey);
  hipStream_t s = thread_stream();
  const size_t words = e->ctx->ct_words(2);
  u64* buf = g_buffers.get(words);
  if (!buf) return from_status(kOutOfMemory);
  CombReq rq;
  rq.kind = 2, rq.in0 = x->dev, rq.key = key, rq.elt = elt, rq.out = buf;
  int st = combine_run(e, rq, s);
  if (st) {
    g_buffers.put(buf, words);
    return from_status(st);
  }
  return finish_result(e, d, 2, buf, words, s, true, rq.nonzero);
}

// SEAL Evaluator::rotate_internal: use the key for `steps` if present, else the NAF decomposition
static long rotate_internal(EvalObj* e, CipherObj* x, int steps, KeysObj* k, CipherObj* d) {
  if (steps == 0) return HIPBFV_S_OK;
  const u32 elt = e->ev->galois_elt_from_step(steps);
  if (!elt) return fail(HIPBFV_E_INVALIDARG, "step count too large");
  if (k->find((elt - 1) >> 1)) return galois_handle(e, x, elt, k, d);
  std::vector<int> naf;
  {
    const bool neg = steps < 0;
    int v = neg ? -steps : steps;
    for (int i = 0; v; i++) {
      const int zi = (v & 1) ? 2 - (v & 3) : 0;
      v = (v - zi) >> 1;
      if (zi) naf.push_back((neg ? -zi : zi) * (1 << i));
    }
  }
  if (naf.size() == 1) return fail(HIPBFV_E_INVALIDARG, "Galois key not present");
  for (int part : naf) {
    if ((u32)(part < 0 ? -part : part) == (e->ctx->n() >> 1)) continue;
    long hr = rotate_internal(e, d, part, k, d);
    if (hr != HIPBFV_S_OK) return hr;
  }
  return HIPBFV_S_OK;
}

static long rotate_common(void* h, void* a, bool columns, int steps, void* keys, void* dst) {
  EvalObj* e = as<EvalObj>(h, kMagicEval);
  CipherObj *x = as<CipherObj>(a, kMagicCipher), *d = as<CipherObj>(dst, kMagicCipher);
  KeysObj* k = as<KeysObj>(keys, kMagicKeys);
  if (!e || !x || !d || !k) return HIPBFV_E_POINTER;
  if (EvalObj* le = level_eval(e, x->ctx)) e = le;
  if (!same_context(x, e)) return fail(HIPBFV_E_INVALIDARG, "encrypted is not valid for encryption parameters");
  if (!e->ctx->batching()) return fail(HIPBFV_COR_E_INVALIDOPERATION, "encryption parameters do not support batching");
  if (x->size != 2) return fail(HIPBFV_E_INVALIDARG, "encrypted size must be 2");
  if (k->ctx && !in_chain(k->ctx, e->ctx.get())) return fail(HIPBFV_E_INVALIDARG, "galois_keys is not valid for encryption parameters");
  if (columns) return galois_handle(e, x, 2 * e->ctx->n() - 1, k, d);
  if (d != x) {  // work on a copy in the destination so that the NAF chain can run in place
    u64* buf = g_buffers.get(x->words);
    if (!buf) return from_status(kOutOfMemory);
    if (copy_d2d(buf, x->dev, x->words * sizeof(u64)) != hipSuccess) {
      g_buffers.put(buf, x->words);
      return from_status(kHipError);
    }
    d->adopt(x->ctx, 2, buf, x->words);
  }
  return rotate_internal(e, d, steps, k, d);
}

long Evaluator_RotateRows(void* h, void* a, int steps, void* keys, void* dst, void* pool) HIPBFV_BEGIN
  (void)pool;
  return rotate_common(h, a, false, steps, keys, dst);
HIPBFV_END

long Evaluator_RotateColumns(void* h, void* a, void* keys, void* dst, void* pool) HIPBFV_BEGIN
  (void)pool;
  return rotate_common(h, a, true, 0, keys, dst);
HIPBFV_END

// ------------------------------------------------------------------ batched device-pointer API
// every hipbfv_batch_* call watches its results for transparent ciphertexts (asynchronously, in the evaluator's status
// word): hipbfv_batch_status reads and resets it
#define EVAL_OR_RETURN(h)                      \
  EvalObj* e = as<EvalObj>(h, kMagicEval);     \
  if (!e) return HIPBFV_E_POINTER;             \
  WatchScope watch_scope_(e->batch_watch && g_throw_transparent ? e->ev->batch_status() : nullptr);

static const u64* key_or_null(void* keys, EvalObj* e, u32 index) {
  KeysObj* k = as<KeysObj>(keys, kMagicKeys);
  if (!k || k->ctx.get() != e->ctx.get()) return nullptr;
  return k->find(index);
}

long hipbfv_batch_multiply(void* h, const uint64_t* a, uint64_t sa, const uint64_t* b, uint64_t sb, uint64_t* out, uint64_t count, void* stream) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  if (!a || !b || !out) return HIPBFV_E_POINTER;
  return from_status(e->ev->multiply((const u64*)a, (u32)sa, (const u64*)b, (u32)sb, (u64*)out, count, (hipStream_t)stream));
HIPBFV_END

long hipbfv_batch_relinearize(void* h, const uint64_t* ct3, void* keys, uint64_t* out2, uint64_t count, void* stream) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  if (!ct3 || !out2) return HIPBFV_E_POINTER;
  const u64* rk = key_or_null(keys, e, 0);
  if (!rk) return from_status(kNoKey);
  return from_status(e->ev->relinearize((const u64*)ct3, rk, (u64*)out2, count, (hipStream_t)stream));
HIPBFV_END

long hipbfv_batch_multiply_relin(void* h, const uint64_t* a, const uint64_t* b, void* keys, uint64_t* out2, uint64_t count, void* stream) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  if (!a || !b || !out2) return HIPBFV_E_POINTER;
  const u64* rk = key_or_null(keys, e, 0);
  if (!rk) return from_status(kNoKey);
  return from_status(e->ev->multiply_relin((const u64*)a, (const u64*)b, rk, (u64*)out2, count, (hipStream_t)stream));
HIPBFV_END

long hipbfv_batch_apply_galois(void* h, const uint64_t* ct2, uint32_t elt, void* keys, uint64_t* out2, uint64_t count, void* stream) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  if (!ct2 || !out2) return HIPBFV_E_POINTER;
  if (!(elt & 1) || elt >= 2 * e->ctx->n()) return from_status(kInvalidArg);
  const u64* key = key_or_null(keys, e, (elt - 1) >> 1);
  if (!key) return from_status(kNoKey);
  return from_status(e->ev->apply_galois((const u64*)ct2, elt, key, (u64*)out2, count, (hipStream_t)stream));
HIPBFV_END

// all items rotate by the same step (each Galois key is streamed once per batch); NAF chain like SEAL
static long batch_rotate_internal(EvalObj* e, const u64* in, int steps, void* keys, u64* out, uint64_t count, hipStream_t s) {
  if (steps == 0) return HIPBFV_S_OK;
  const u32 elt = e->ev->galois_elt_from_step(steps);
  if (!elt) return fail(HIPBFV_E_INVALIDARG, "step count too large");
  if (const u64* key = key_or_null(keys, e, (elt - 1) >> 1))
    return from_status(e->ev->apply_galois(in, elt, key, out, count, s));
  std::vector<int> naf;
  const bool neg = steps < 0;
  int v = neg ? -steps : steps;
  for (int i = 0; v; i++) {
    const int zi = (v & 1) ? 2 - (v & 3) : 0;
    v = (v - zi) >> 1;
    if (zi) naf.push_back((neg ? -zi : zi) * (1 << i));
  }
  if (naf.size() == 1) return from_status(kNoKey);
  const u64* cur = in;
  for (int part : naf) {
    if ((u32)(part < 0 ? -part : part) == (e->ctx->n() >> 1)) continue;
    long hr = batch_rotate_internal(e, cur, part, keys, out, count, s);
    if (hr != HIPBFV_S_OK) return hr;
    cur = out;
  }
  return HIPBFV_S_OK;
}

long hipbfv_batch_rotate_rows(void* h, const uint64_t* ct2, int steps, void* keys, uint64_t* out2, uint64_t count, void* stream) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  if (!ct2 || !out2) return HIPBFV_E_POINTER;
  if (!e->ctx->batching()) return fail(HIPBFV_COR_E_INVALIDOPERATION, "encryption parameters do not support batching");
  hipStream_t s = (hipStream_t)stream;
  if (steps == 0) {
    if ((const u64*)ct2 != (u64*)out2 &&
        hipMemcpyAsync(out2, ct2, count * e->ctx->ct_words(2) * sizeof(u64), hipMemcpyDeviceToDevice, s) != hipSuccess)
      return from_status(kHipError);
    return HIPBFV_S_OK;
  }
  return batch_rotate_internal(e, (const u64*)ct2, steps, keys, (u64*)out2, count, s);
HIPBFV_END

long hipbfv_batch_rotate_columns(void* h, const uint64_t* ct2, void* keys, uint64_t* out2, uint64_t count, void* stream) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  if (!e->ctx->batching()) return fail(HIPBFV_COR_E_INVALIDOPERATION, "encryption parameters do not support batching");
  return hipbfv_batch_apply_galois(h, ct2, 2 * e->ctx->n() - 1, keys, out2, count, stream);
HIPBFV_END

// ---- per-key batches: item i of the batch uses key set key_index[i] (include/hipbfv.h) ----
// the key `index` (0 = relinearisation, (elt - 1) / 2 = Galois) of every set, as the evaluator's per-item selection; `tab` keeps
// the pointer table alive for the call.  !present(): a handle is not a key object of this context, or a set lacks the key.
static KeySel keys_sel(void* const* key_sets, uint64_t num_sets, const uint32_t* key_index, uint64_t count, EvalObj* e, u32 index,
                       std::vector<const u64*>& tab) {
  tab.assign(num_sets, nullptr);
  for (uint64_t k = 0; k < num_sets; k++)
    if (!(tab[k] = key_or_null(key_sets[k], e, index))) return KeySel();
  KeySel sel;
  sel.keys = tab.data();
  sel.nkeys = (u32)num_sets;
  sel.index = key_index;
  sel.period = (size_t)count;
  return sel;
}
#define KEYSETS_OR_RETURN()                                                                   \
  if (!key_sets || !key_index || !num_sets || num_sets > 0xFFFFFFFFull) return HIPBFV_E_POINTER; \
  for (uint64_t i__ = 0; i__ < count; i__++)                                                  \
    if (key_index[i__] >= num_sets) return fail(HIPBFV_E_INVALIDARG, "key_index names a key set that was not given");

long hipbfv_batch_relinearize_keys(void* h, const uint64_t* ct3, void* const* key_sets, uint64_t num_sets, const uint32_t* key_index,
                                   uint64_t* out2, uint64_t count, void* stream) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  if (!ct3 || !out2) return HIPBFV_E_POINTER;
  KEYSETS_OR_RETURN();
  if (!count) return HIPBFV_S_OK;
  std::vector<const u64*> tab;
  const KeySel sel = keys_sel(key_sets, num_sets, key_index, count, e, 0, tab);
  if (!sel.present()) return from_status(kNoKey);
  return from_status(e->ev->relinearize((const u64*)ct3, sel, (u64*)out2, count, (hipStream_t)stream));
HIPBFV_END

long hipbfv_batch_multiply_relin_keys(void* h, const uint64_t* a, const uint64_t* b, void* const* key_sets, uint64_t num_sets,
                                      const uint32_t* key_index, uint64_t* out2, uint64_t count, void* stream) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  if (!a || !b || !out2) return HIPBFV_E_POINTER;
  KEYSETS_OR_RETURN();
  if (!count) return HIPBFV_S_OK;
  std::vector<const u64*> tab;
  const KeySel sel = keys_sel(key_sets, num_sets, key_index, count, e, 0, tab);
  if (!sel.present()) return from_status(kNoKey);
  return from_status(e->ev->multiply_relin((const u64*)a, (const u64*)b, sel, (u64*)out2, count, (hipStream_t)stream));
HIPBFV_END

long hipbfv_batch_apply_galois_keys(void* h, const uint64_t* ct2, uint32_t elt, void* const* key_sets, uint64_t num_sets,
                                    const uint32_t* key_index, uint64_t* out2, uint64_t count, void* stream) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  if (!ct2 || !out2) return HIPBFV_E_POINTER;
  if (!(elt & 1) || elt >= 2 * e->ctx->n()) return from_status(kInvalidArg);
  KEYSETS_OR_RETURN();
  if (!count) return HIPBFV_S_OK;
  std::vector<const u64*> tab;
  const KeySel sel = keys_sel(key_sets, num_sets, key_index, count, e, (elt - 1) >> 1, tab);
  if (!sel.present()) return from_status(kNoKey);
  return from_status(e->ev->apply_galois((const u64*)ct2, elt, sel, (u64*)out2, count, (hipStream_t)stream));
HIPBFV_END

// SEAL's rotate_internal over per-item key sets: the direct key when EVERY set has it, the NAF chain otherwise
static long batch_rotate_keys_internal(EvalObj* e, const u64* in, int steps, void* const* key_sets, uint64_t num_sets, const uint32_t* key_index,
                                       u64* out, uint64_t count, hipStream_t s) {
  if (steps == 0) return HIPBFV_S_OK;
  const u32 elt = e->ev->galois_elt_from_step(steps);
  if (!elt) return fail(HIPBFV_E_INVALIDARG, "step count too large");
  std::vector<const u64*> tab;
  if (const KeySel sel = keys_sel(key_sets, num_sets, key_index, count, e, (elt - 1) >> 1, tab); sel.present())
    return from_status(e->ev->apply_galois(in, elt, sel, out, count, s));
  std::vector<int> naf;
  const bool neg = steps < 0;
  int v = neg ? -steps : steps;
  for (int i = 0; v; i++) {
    const int zi = (v & 1) ? 2 - (v & 3) : 0;
    v = (v - zi) >> 1;
    if (zi) naf.push_back((neg ? -zi : zi) * (1 << i));
  }
  if (naf.size() == 1) return from_status(kNoKey);
  const u64* cur = in;
  for (int part : naf) {
    if ((u32)(part < 0 ? -part : part) == (e->ctx->n() >> 1)) continue;
    long hr = batch_rotate_keys_internal(e, cur, part, key_sets, num_sets, key_index, out, count, s);
    if (hr != HIPBFV_S_OK) return hr;
    cur = out;
  }
  return HIPBFV_S_OK;
}

long hipbfv_batch_rotate_rows_keys(void* h, const uint64_t* ct2, int steps, void* const* key_sets, uint64_t num_sets, const uint32_t* key_index,
                                   uint64_t* out2, uint64_t count, void* stream) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  if (!ct2 || !out2) return HIPBFV_E_POINTER;
  if (!e->ctx->batching()) return fail(HIPBFV_COR_E_INVALIDOPERATION, "encryption parameters do not support batching");
  KEYSETS_OR_RETURN();
  hipStream_t s = (hipStream_t)stream;
  if (steps == 0) {
    if ((const u64*)ct2 != (u64*)out2 &&
        hipMemcpyAsync(out2, ct2, count * e->ctx->ct_words(2) * sizeof(u64), hipMemcpyDeviceToDevice, s) != hipSuccess)
      return from_status(kHipError);
    return HIPBFV_S_OK;
  }
  if (!count) return HIPBFV_S_OK;
  return batch_rotate_keys_internal(e, (const u64*)ct2, steps, key_sets, num_sets, key_index, (u64*)out2, count, s);
HIPBFV_END

long hipbfv_batch_rotate_columns_keys(void* h, const uint64_t* ct2, void* const* key_sets, uint64_t num_sets, const uint32_t* key_index,
                                      uint64_t* out2, uint64_t count, void* stream) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  if (!e->ctx->batching()) return fail(HIPBFV_COR_E_INVALIDOPERATION, "encryption parameters do not support batching");
  return hipbfv_batch_apply_galois_keys(h, ct2, 2 * e->ctx->n() - 1, key_sets, num_sets, key_index, out2, count, stream);
HIPBFV_END

long hipbfv_batch_add(void* h, const uint64_t* a, const uint64_t* b, uint64_t* out, uint64_t size, uint64_t count, void* stream) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  if (!a || !b || !out) return HIPBFV_E_POINTER;
  return from_status(e->ev->add((const u64*)a, (const u64*)b, (u64*)out, (u32)size, count, (hipStream_t)stream));
HIPBFV_END

long hipbfv_batch_sub(void* h, const uint64_t* a, const uint64_t* b, uint64_t* out, uint64_t size, uint64_t count, void* stream) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  if (!a || !b || !out) return HIPBFV_E_POINTER;
  return from_status(e->ev->sub((const u64*)a, (const u64*)b, (u64*)out, (u32)size, count, (hipStream_t)stream));
HIPBFV_END

long hipbfv_batch_negate(void* h, const uint64_t* a, uint64_t* out, uint64_t size, uint64_t count, void* stream) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  if (!a || !out) return HIPBFV_E_POINTER;
  return from_status(e->ev->negate((const u64*)a, (u64*)out, (u32)size, count, (hipStream_t)stream));
HIPBFV_END

long hipbfv_batch_add_plain(void* h, const uint64_t* ct, uint64_t size, const uint64_t* plain, uint64_t pstride, uint64_t* out, uint64_t count, void* stream) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  if (!ct || !plain || !out) return HIPBFV_E_POINTER;
  return from_status(e->ev->add_plain((const u64*)ct, (u32)size, (const u64*)plain, pstride, (u64*)out, count, (hipStream_t)stream));
HIPBFV_END

long hipbfv_batch_sub_plain(void* h, const uint64_t* ct, uint64_t size, const uint64_t* plain, uint64_t pstride, uint64_t* out, uint64_t count, void* stream) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  if (!ct || !plain || !out) return HIPBFV_E_POINTER;
  return from_status(e->ev->sub_plain((const u64*)ct, (u32)size, (const u64*)plain, pstride, (u64*)out, count, (hipStream_t)stream));
HIPBFV_END

long hipbfv_batch_multiply_plain(void* h, const uint64_t* ct, uint64_t size, const uint64_t* plain, uint64_t pstride, uint64_t* out, uint64_t count, void* stream) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  if (!ct || !plain || !out) return HIPBFV_E_POINTER;
  return from_status(e->ev->multiply_plain((const u64*)ct, (u32)size, (const u64*)plain, pstride, (u64*)out, count, (hipStream_t)stream));
HIPBFV_END

long hipbfv_batch_ntt(void* h, uint64_t* data, uint64_t polys, uint64_t nprimes, bool inverse, void* stream) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  if (!data) return HIPBFV_E_POINTER;
  return from_status(e->ev->ntt((u64*)data, polys, (u32)nprimes, inverse, (hipStream_t)stream));
HIPBFV_END

// ------------------------------------------------------------------ SEAL 4.0 wire format (wire.cpp)
// Ceilings on a decompressed object body for a given context (the bytes are untrusted: a small zstd frame may declare any
// content size).  SEAL_CIPHERTEXT_SIZE_MAX = 16 polynomials; a key-switching key set holds at most one key per odd Galois
// element actually stored -- 8 * log2(N) keys is four times SEAL's default set.
static size_t max_ct_body(const Context& c) { return 4096 + (size_t)16 * c.n() * c.KK() * 8; }
static size_t max_plain_body(const Context& c) { return 4096 + (size_t)c.n() * c.KK() * 8; }
static size_t max_keys_body(const Context& c) {
  size_t lg = 0;
  while (((size_t)1 << lg) < c.n()) lg++;
  return 4096 + 8 * (size_t)c.n() + 8 * lg * c.K() * (512 + (size_t)2 * c.KK() * c.n() * 8);
}

static long from_wire(int rc) {
  switch (rc) {
    case kWireOk: return HIPBFV_S_OK;
    case kWireBadArg: return fail(HIPBFV_E_INVALIDARG, "unsupported compression mode");
    case kWireNoZstd: return fail(HIPBFV_COR_E_IO, "libzstd.so.1 is not available: only compr_mode 0 (none) can be used");
    case kWireSeeded:
      return fail(HIPBFV_COR_E_IO, "seed-compressed (compact) SEAL object: expanding SEAL's PRNG stream is not supported, save the object without save_seed");
    default: return fail(HIPBFV_COR_E_IO, "malformed or truncated SEAL object");
  }
}

static void data_level_parms_id(const Context& c, uint8_t out[32]) { seal_parms_id(c.n(), c.key_primes().data(), c.K(), c.t(), out); }
static void key_level_parms_id(const Context& c, uint8_t out[32]) { seal_parms_id(c.n(), c.key_primes().data(), c.KK(), c.t(), out); }

long hipbfv_wire_parms_id(uint64_t n, const uint64_t* primes, uint64_t count, uint64_t plain_modulus, uint8_t* out32) HIPBFV_BEGIN
  if (!primes || !out32) return HIPBFV_E_POINTER;
  seal_parms_id(n, reinterpret_cast<const unsigned long long*>(primes), count, plain_modulus, out32);
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_wire_decode_ciphertext(const uint8_t* in, uint64_t in_size, uint8_t* parms_id32, bool* is_ntt, uint64_t* size, uint64_t* n,
                                   uint64_t* k, uint64_t* data, uint64_t capacity_words, int64_t* in_bytes) HIPBFV_BEGIN
  if (!in) return HIPBFV_E_POINTER;
  WireCiphertext ct;
  size_t used = 0;
  if (int rc = wire_unpack_ciphertext(in, in_size, &ct, &used, data ? 4096 + (size_t)capacity_words * 8 : 0)) return from_wire(rc);
  if (parms_id32) std::memcpy(parms_id32, ct.parms_id, 32);
  if (is_ntt) *is_ntt = ct.is_ntt;
  if (size) *size = ct.size;
  if (n) *n = ct.n;
  if (k) *k = ct.k;
  if (in_bytes) *in_bytes = (int64_t)used;
  if (data) {
    if (capacity_words < ct.data.size()) return fail(HIPBFV_E_INVALIDARG, "buffer too small");
    std::memcpy(data, ct.data.data(), ct.data.size() * 8);
  }
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_wire_encode_ciphertext(const uint8_t* parms_id32, bool is_ntt, uint64_t size, uint64_t n, uint64_t k, const uint64_t* data,
                                   uint8_t compr_mode, uint8_t* out, uint64_t capacity, int64_t* out_bytes) HIPBFV_BEGIN
  if (!parms_id32 || !data || !out_bytes) return HIPBFV_E_POINTER;
  std::vector<uint8_t> buf;
  if (int rc = wire_pack_ciphertext(parms_id32, is_ntt, size, n, k, reinterpret_cast<const unsigned long long*>(data), compr_mode, &buf)) return from_wire(rc);
  *out_bytes = (int64_t)buf.size();
  if (!out) return HIPBFV_S_OK;  // size query
  if (capacity < buf.size()) return fail(HIPBFV_E_INVALIDARG, "buffer too small");
  std::memcpy(out, buf.data(), buf.size());
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_wire_decode_plaintext(const uint8_t* in, uint64_t in_size, uint8_t* parms_id32, uint64_t* coeff_count, uint64_t* coeffs,
                                  uint64_t capacity_words, int64_t* in_bytes) HIPBFV_BEGIN
  if (!in) return HIPBFV_E_POINTER;
  WirePlaintext pt;
  size_t used = 0;
  if (int rc = wire_unpack_plaintext(in, in_size, &pt, &used, coeffs ? 4096 + (size_t)capacity_words * 8 : 0)) return from_wire(rc);
  if (parms_id32) std::memcpy(parms_id32, pt.parms_id, 32);
  if (coeff_count) *coeff_count = pt.coeffs.size();
  if (in_bytes) *in_bytes = (int64_t)used;
  if (coeffs) {
    if (capacity_words < pt.coeffs.size()) return fail(HIPBFV_E_INVALIDARG, "buffer too small");
    std::memcpy(coeffs, pt.coeffs.data(), pt.coeffs.size() * 8);
  }
  return HIPBFV_S_OK;
HIPBFV_END

// upper bound of the serialised size, like SEAL's save_size()
long Ciphertext_SaveSize(void* h, uint8_t compr_mode, int64_t* result) HIPBFV_BEGIN
  CipherObj* c = as<CipherObj>(h, kMagicCipher);
  if (!c || !result) return HIPBFV_E_POINTER;
  const size_t raw = 16 + 32 + 1 + 8 * 5 + 16 + 8 + c->words * 8;
  *result = (int64_t)(compr_mode ? raw + raw / 128 + 512 : raw);
  return HIPBFV_S_OK;
HIPBFV_END

long Ciphertext_Save(void* h, uint8_t* outptr, uint64_t size, uint8_t compr_mode, int64_t* out_bytes) HIPBFV_BEGIN
  CipherObj* c = as<CipherObj>(h, kMagicCipher);
  if (!c || !outptr || !out_bytes) return HIPBFV_E_POINTER;
  if (!c->ctx || !c->dev) return fail(HIPBFV_E_INVALIDARG, "ciphertext is empty");
  long hr = ensure_host(c);
  if (hr != HIPBFV_S_OK) return hr;
  uint8_t pid[32];
  data_level_parms_id(*c->ctx, pid);
  std::vector<uint8_t> buf;
  if (int rc = wire_pack_ciphertext(pid, false, c->size, c->ctx->n(), c->ctx->K(), c->host.data(), compr_mode, &buf)) return from_wire(rc);
  if (size < buf.size()) return fail(HIPBFV_E_INVALIDARG, "buffer too small");
  std::memcpy(outptr, buf.data(), buf.size());
  *out_bytes = (int64_t)buf.size();
  return HIPBFV_S_OK;
HIPBFV_END

long Ciphertext_Load(void* h, void* context, uint8_t* inptr, uint64_t size, int64_t* in_bytes) HIPBFV_BEGIN
  CipherObj* c = as<CipherObj>(h, kMagicCipher);
  ContextObj* x = as<ContextObj>(context, kMagicContext);
  if (!c || !x || !inptr || !in_bytes) return HIPBFV_E_POINTER;
  WireCiphertext ct;
  size_t used = 0;
  if (int rc = wire_unpack_ciphertext(inptr, size, &ct, &used, max_ct_body(*x->ctx))) return from_wire(rc);
  // the parms_id names the level of the modulus-switching chain the ciphertext lives at
  std::shared_ptr<Context> lvl = x->ctx;
  for (;;) {
    uint8_t pid[32];
    data_level_parms_id(*lvl, pid);
    if (std::memcmp(pid, ct.parms_id, 32) == 0) break;
    lvl = ct.k < lvl->K() ? lvl->next_level(nullptr) : nullptr;
    if (!lvl) return fail(HIPBFV_E_INVALIDARG, "ciphertext data is invalid for the encryption parameters");
  }
  if (ct.is_ntt || ct.n != lvl->n() || ct.k != lvl->K() || ct.size < 2)
    return fail(HIPBFV_E_INVALIDARG, "ciphertext data is invalid for the encryption parameters");
  ContextObj level_handle;
  level_handle.ctx = lvl;
  long hr = hipbfv_Ciphertext_Assign(h, &level_handle, ct.size, reinterpret_cast<const uint64_t*>(ct.data.data()));
  if (hr != HIPBFV_S_OK) return hr;
  *in_bytes = (int64_t)used;
  return HIPBFV_S_OK;
HIPBFV_END

long Plaintext_SaveSize(void* h, uint8_t compr_mode, int64_t* result) HIPBFV_BEGIN
  PlainObj* p = as<PlainObj>(h, kMagicPlain);
  if (!p || !result) return HIPBFV_E_POINTER;
  const size_t raw = 16 + 32 + 16 + 16 + 8 + p->coeffs.size() * 8;
  *result = (int64_t)(compr_mode ? raw + raw / 128 + 512 : raw);
  return HIPBFV_S_OK;
HIPBFV_END

long Plaintext_Save(void* h, uint8_t* outptr, uint64_t size, uint8_t compr_mode, int64_t* out_bytes) HIPBFV_BEGIN
  PlainObj* p = as<PlainObj>(h, kMagicPlain);
  if (!p || !outptr || !out_bytes) return HIPBFV_E_POINTER;
  const uint8_t zero[32] = {0};  // BFV plaintexts in coefficient form carry parms_id_zero
  std::vector<uint8_t> buf;
  if (int rc = wire_pack_plaintext(zero, p->coeffs.data(), p->coeffs.size(), compr_mode, &buf)) return from_wire(rc);
  if (size < buf.size()) return fail(HIPBFV_E_INVALIDARG, "buffer too small");
  std::memcpy(outptr, buf.data(), buf.size());
  *out_bytes = (int64_t)buf.size();
  return HIPBFV_S_OK;
HIPBFV_END

long Plaintext_Load(void* h, void* context, uint8_t* inptr, uint64_t size, int64_t* in_bytes) HIPBFV_BEGIN
  PlainObj* p = as<PlainObj>(h, kMagicPlain);
  ContextObj* x = as<ContextObj>(context, kMagicContext);
  if (!p || !x || !inptr || !in_bytes) return HIPBFV_E_POINTER;
  WirePlaintext pt;
  size_t used = 0;
  if (int rc = wire_unpack_plaintext(inptr, size, &pt, &used, max_plain_body(*x->ctx))) return from_wire(rc);
  const uint8_t zero[32] = {0};
  if (std::memcmp(pt.parms_id, zero, 32) != 0) return fail(HIPBFV_E_INVALIDARG, "NTT-form plaintexts are not used by BFV evaluation");
  if (pt.coeffs.size() > x->ctx->n()) return fail(HIPBFV_E_INVALIDARG, "plaintext data is invalid for the encryption parameters");
  for (u64 v : pt.coeffs)
    if (v >= x->ctx->t()) return fail(HIPBFV_E_INVALIDARG, "plaintext data is invalid for the encryption parameters");
  p->coeffs = pt.coeffs;
  *in_bytes = (int64_t)used;
  return HIPBFV_S_OK;
HIPBFV_END

static long keys_to_host(KeysObj* k, std::vector<std::vector<u64>>* host, std::vector<std::vector<const u64*>>* ptrs) {
  const size_t KK = k->ctx->KK(), K = k->ctx->K(), n = k->ctx->n();
  u32 max_index = 0;
  for (auto& kv : k->keys) max_index = std::max(max_index, kv.first);
  ptrs->assign(k->keys.empty() ? 0 : max_index + 1, {});
  for (auto& kv : k->keys) {
    host->emplace_back(k->ctx->key_words());
    std::vector<u64>& buf = host->back();
    if (hipMemcpy(buf.data(), kv.second, buf.size() * sizeof(u64), hipMemcpyDeviceToHost) != hipSuccess) return from_status(kHipError);
    for (size_t J = 0; J < K; J++) (*ptrs)[kv.first].push_back(buf.data() + J * 2 * KK * n);
  }
  return HIPBFV_S_OK;
}

long KSwitchKeys_SaveSize(void* h, uint8_t compr_mode, int64_t* result) HIPBFV_BEGIN
  KeysObj* k = as<KeysObj>(h, kMagicKeys);
  if (!k || !result) return HIPBFV_E_POINTER;
  size_t raw = 16 + 32 + 8;
  if (k->ctx) {
    u32 max_index = 0;
    for (auto& kv : k->keys) max_index = std::max(max_index, kv.first);
    raw += 8 * (size_t)(max_index + 1);
    raw += k->keys.size() * k->ctx->K() * (16 + 32 + 1 + 40 + 24 + 2 * k->ctx->KK() * k->ctx->n() * 8);
  }
  *result = (int64_t)(compr_mode ? raw + raw / 128 + 512 : raw);
  return HIPBFV_S_OK;
HIPBFV_END

long KSwitchKeys_Save(void* h, uint8_t* outptr, uint64_t size, uint8_t compr_mode, int64_t* out_bytes) HIPBFV_BEGIN
  KeysObj* k = as<KeysObj>(h, kMagicKeys);
  if (!k || !outptr || !out_bytes) return HIPBFV_E_POINTER;
  if (!k->ctx) return fail(HIPBFV_E_INVALIDARG, "keys are empty");
  std::vector<std::vector<u64>> host;
  host.reserve(k->keys.size());
  std::vector<std::vector<const u64*>> ptrs;
  long hr = keys_to_host(k, &host, &ptrs);
  if (hr != HIPBFV_S_OK) return hr;
  uint8_t pid[32];
  key_level_parms_id(*k->ctx, pid);
  std::vector<uint8_t> buf;
  if (int rc = wire_pack_kswitch(pid, k->ctx->n(), k->ctx->KK(), ptrs, compr_mode, &buf)) return from_wire(rc);
  if (size < buf.size()) return fail(HIPBFV_E_INVALIDARG, "buffer too small");
  std::memcpy(outptr, buf.data(), buf.size());
  *out_bytes = (int64_t)buf.size();
  return HIPBFV_S_OK;
HIPBFV_END

long KSwitchKeys_Load(void* h, void* context, uint8_t* inptr, uint64_t size, int64_t* in_bytes) HIPBFV_BEGIN
  KeysObj* k = as<KeysObj>(h, kMagicKeys);
  ContextObj* x = as<ContextObj>(context, kMagicContext);
  if (!k || !x || !inptr || !in_bytes) return HIPBFV_E_POINTER;
  WireKSwitchKeys ks;
  size_t used = 0;
  if (int rc = wire_unpack_kswitch(inptr, size, &ks, &used, max_keys_body(*x->ctx), x->ctx->n())) return from_wire(rc);
  uint8_t pid[32];
  key_level_parms_id(*x->ctx, pid);
  if (std::memcmp(pid, ks.parms_id, 32) != 0) return fail(HIPBFV_E_INVALIDARG, "keys are invalid for the encryption parameters");
  const size_t KK = x->ctx->KK(), K = x->ctx->K(), n = x->ctx->n();
  std::vector<u64> flat(x->ctx->key_words());
  for (size_t index = 0; index < ks.keys.size(); index++) {
    const auto& entry = ks.keys[index];
    if (entry.empty()) continue;
    if (entry.size() != K) return fail(HIPBFV_E_INVALIDARG, "keys are invalid for the encryption parameters");
    for (size_t J = 0; J < K; J++) {
      const WireCiphertext& pk = entry[J];
      if (!pk.is_ntt || pk.size != 2 || pk.n != n || pk.k != KK) return fail(HIPBFV_E_INVALIDARG, "keys are invalid for the encryption parameters");
      std::memcpy(flat.data() + J * 2 * KK * n, pk.data.data(), 2 * KK * n * 8);
    }
    long hr = assign_key(k, x, (u32)index, reinterpret_cast<const uint64_t*>(flat.data()));
    if (hr != HIPBFV_S_OK) return hr;
  }
  *in_bytes = (int64_t)used;
  return HIPBFV_S_OK;
HIPBFV_END

// ------------------------------------------------------------------ program graphs (batch executor)
long hipbfv_Program_Create(void** out) HIPBFV_BEGIN
  if (!out) return HIPBFV_E_POINTER;
  *out = new ProgramObj();
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_Program_Destroy(void* h) HIPBFV_BEGIN
  ProgramObj* p = as<ProgramObj>(h, kMagicProgram);
  if (!p) return HIPBFV_E_POINTER;
  delete p;
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_Program_AddNode(void* h, uint32_t op, uint64_t arg, uint32_t* node_id) HIPBFV_BEGIN
  ProgramObj* p = as<ProgramObj>(h, kMagicProgram);
  if (!p || !node_id) return HIPBFV_E_POINTER;
  if (op >= (uint32_t)kOpCount || op == (uint32_t)kOpLiteralPlaintext)
    return fail(HIPBFV_E_INVALIDARG, "unknown operation kind (plaintext literals: hipbfv_Program_AddPlaintextLiteral)");
  *node_id = (uint32_t)p->prog.add_node((OpKind)op, arg);
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_Program_AddPlaintextLiteral(void* h, const uint8_t* bytes, uint64_t length, uint32_t* node_id) HIPBFV_BEGIN
  ProgramObj* p = as<ProgramObj>(h, kMagicProgram);
  if (!p || !bytes || !node_id) return HIPBFV_E_POINTER;
  std::string err;
  const int id = p->prog.add_plaintext_literal(bytes, (size_t)length, &err);
  if (id < 0) return fail(HIPBFV_E_INVALIDARG, err.c_str());
  *node_id = (uint32_t)id;
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_Program_AddEdge(void* h, uint32_t src, uint32_t dst, uint32_t kind) HIPBFV_BEGIN
  ProgramObj* p = as<ProgramObj>(h, kMagicProgram);
  if (!p) return HIPBFV_E_POINTER;
  if (kind > 2 || p->prog.add_edge((int)src, (int)dst, (EdgeKind)kind) != kOk) return fail(HIPBFV_E_INVALIDARG, "invalid edge");
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_Program_LoadJson(void* h, const char* json, uint64_t length) HIPBFV_BEGIN
  ProgramObj* p = as<ProgramObj>(h, kMagicProgram);
  if (!p || !json) return HIPBFV_E_POINTER;
  std::string err;
  if (p->prog.load_json(json, length, &err) != kOk) return fail(HIPBFV_E_INVALIDARG, err.c_str());
  if (p->prog.validate(&err) != kOk) return fail(HIPBFV_E_INVALIDARG, err.c_str());
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_Program_NumOutputs(void* h, uint64_t* count) HIPBFV_BEGIN
  ProgramObj* p = as<ProgramObj>(h, kMagicProgram);
  if (!p || !count) return HIPBFV_E_POINTER;
  *count = p->prog.num_outputs();
  return HIPBFV_S_OK;
HIPBFV_END

// Diagnostic, host only: the FP64 range plans of one prime (context.cpp); the CPU suite replays them against an independent
// worst-case model of the butterfly arithmetic.
long hipbfv_debug_f64_plan(uint64_t prime, uint32_t log_n, uint32_t* out6) HIPBFV_BEGIN
  if (!out6) return fail(HIPBFV_E_POINTER, "null output");
  if (log_n < 10 || log_n > 15) return fail(HIPBFV_E_INVALIDARG, "log_n outside 10..15");
  u32 out[6];
  debug_f64_plan(prime, (int)log_n, out);
  for (int i = 0; i < 6; i++) out6[i] = out[i];
  return HIPBFV_S_OK;
HIPBFV_END

// Diagnostic, host only: the auxiliary base Context::create would pick for these parameters (no device is touched: the context
// is built with device = -1 and dropped).  out = {B_1 .. B_nB, m_sk}; flags as hipbfv_Context_AuxBase.
long hipbfv_debug_aux_base(uint64_t poly_modulus_degree, const uint64_t* coeff_primes, uint64_t prime_count, uint64_t plain_modulus,
                           uint64_t* count, uint64_t* primes, uint64_t capacity, int* flags) HIPBFV_BEGIN
  if (!coeff_primes || !count) return HIPBFV_E_POINTER;
  std::string err;
  std::unique_ptr<hipbfv::Context> c(hipbfv::Context::create((u32)poly_modulus_degree, std::vector<u64>(coeff_primes, coeff_primes + prime_count), plain_modulus, -1, &err));
  if (!c) return fail(HIPBFV_E_INVALIDARG, err.c_str());
  const hipbfv::DevCtx& d = c->host();
  *count = d.S;
  if (flags) *flags = (d.aux_f64 ? 1 : 0) | (d.pack_mul ? 2 : 0) | (d.pack_ks ? 4 : 0) | (d.conv_grid ? 8 : 0) | (d.aux_mixed ? 16 : 0) | (d.pack_mul == 2 ? 32 : 0) | (d.pack_ks == 2 ? 64 : 0);
  if (primes) {
    if (capacity < d.S) return fail(HIPBFV_E_INVALIDARG, "capacity too small");
    for (uint32_t j = 0; j < d.S; j++) primes[j] = d.mod[d.KK + j].q;
  }
  return HIPBFV_S_OK;
HIPBFV_END

// Diagnostic (tools/graph_probe.py): one multiply + relinearize of a single ciphertext pair, (1) launched kernel by kernel as
// the handle-level calls do, (2) the same launches captured once into a hipGraph and replayed.  Both are timed from the host
// with one stream synchronisation per repetition, i.e. what a caller of the SEAL-named entry points waits for.
long hipbfv_debug_graph_probe(void* context, const uint64_t* a, const uint64_t* b, const uint64_t* relin_key, uint64_t* out, uint64_t iterations,
                              double* us_direct, double* us_graph) HIPBFV_BEGIN
  ContextObj* c = as<ContextObj>(context, kMagicContext);
  if (!c || !a || !b || !relin_key || !out || !us_direct || !us_graph || !iterations) return HIPBFV_E_POINTER;
  Evaluator ev(c->ctx.get());  // its own scratch pool: events recorded during capture never meet another stream
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return from_status(kHipError);
  const size_t w3 = c->ctx->ct_words(3);
  u64* tmp = nullptr;
  if (hipMalloc((void**)&tmp, w3 * sizeof(u64)) != hipSuccess) {
    (void)hipStreamDestroy(s);
    return from_status(kOutOfMemory);
  }
  auto once = [&]() -> int {
    int st = ev.multiply((const u64*)a, 2, (const u64*)b, 2, tmp, 1, s, false);
    if (!st) st = ev.relinearize(tmp, (const u64*)relin_key, (u64*)out, 1, s, nullptr, false);
    return st;
  };
  auto now_us = [] {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
  };
  long hr = HIPBFV_S_OK;
  int st = once();  // warm: the scratch pool allocates here, not inside the capture
  if (!st && hipStreamSynchronize(s) != hipSuccess) st = kHipError;
  if (!st) {
    const double t0 = now_us();
    for (uint64_t i = 0; i < iterations && !st; i++) {
      st = once();
      if (!st && hipStreamSynchronize(s) != hipSuccess) st = kHipError;
    }
    *us_direct = (now_us() - t0) / (double)iterations;
  }
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  if (!st) {
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) st = kHipError;
    if (!st) st = once();
    if (hipStreamEndCapture(s, &graph) != hipSuccess || !graph) st = st ? st : (int)kHipError;
    if (!st && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) st = kHipError;
  }
  if (!st) {
    for (int i = 0; i < 3 && !st; i++)
      if (hipGraphLaunch(exec, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) st = kHipError;
    const double t0 = now_us();
    for (uint64_t i = 0; i < iterations && !st; i++)
      if (hipGraphLaunch(exec, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) st = kHipError;
    *us_graph = (now_us() - t0) / (double)iterations;
  }
  if (st) hr = from_status(st);
  if (exec) (void)hipGraphExecDestroy(exec);
  if (graph) (void)hipGraphDestroy(graph);
  (void)hipStreamSynchronize(s);
  (void)hipFree(tmp);
  (void)hipStreamDestroy(s);
  return hr;
HIPBFV_END

long hipbfv_Program_Describe(void* h, char* buffer, uint64_t capacity, uint64_t* needed) HIPBFV_BEGIN
  ProgramObj* p = as<ProgramObj>(h, kMagicProgram);
  if (!p || !needed) return HIPBFV_E_POINTER;
  std::string text;
  const int st = p->prog.describe(&text);
  *needed = text.size() + 1;
  if (buffer && capacity) {
    const size_t c = std::min<size_t>(text.size(), capacity - 1);
    std::memcpy(buffer, text.data(), c);
    buffer[c] = 0;
  }
  if (st != kOk) {
    const long hr = from_status(st);
    tls_error = text;  // the schedule's own message ("error: left operand is not a ciphertext")
    return hr;
  }
  return HIPBFV_S_OK;
HIPBFV_END

// key_index == nullptr: one key set (num_key_sets = 1) for every input set -- the reference's call
static long program_run_impl(void* h, void* evaluator, uint64_t batch, uint64_t num_inputs, const uint32_t* input_kinds,
                             const uint64_t* const* input_ptrs, const uint64_t* input_strides, uint64_t num_key_sets, void* const* relin_keys,
                             void* const* galois_keys, const uint32_t* key_index, uint64_t num_outputs, uint64_t* const* outputs, void* stream) {
  ProgramObj* p = as<ProgramObj>(h, kMagicProgram);
  EvalObj* e = as<EvalObj>(evaluator, kMagicEval);
  if (!p || !e || (num_inputs && (!input_kinds || !input_ptrs || !input_strides)) || (num_outputs && !outputs)) return HIPBFV_E_POINTER;
  std::vector<ProgramInput> ins(num_inputs);
  for (uint64_t i = 0; i < num_inputs; i++) ins[i] = ProgramInput{(int)input_kinds[i], (const u64*)input_ptrs[i], (size_t)input_strides[i]};
  ProgramKeys keys;
  keys.relin.assign(num_key_sets, nullptr);
  keys.galois.resize(num_key_sets);
  for (uint64_t k = 0; k < num_key_sets; k++) {
    if (KeysObj* ko = as<KeysObj>(relin_keys ? relin_keys[k] : nullptr, kMagicKeys))
      if (ko->ctx.get() == e->ctx.get()) keys.relin[k] = ko->find(0);
    if (KeysObj* ko = as<KeysObj>(galois_keys ? galois_keys[k] : nullptr, kMagicKeys))
      if (ko->ctx.get() == e->ctx.get())
        for (auto& kv : ko->keys) keys.galois[k][kv.first] = kv.second;
  }
  if (key_index)
    for (uint64_t i = 0; i < batch; i++)
      if (key_index[i] >= num_key_sets) return fail(HIPBFV_E_INVALIDARG, "key_index names a key set that was not given");
  std::string err;
  // the reference's runtime.run fails when any node's result is transparent (SEAL built with throw-on-transparent,
  // seal_fhe/build.rs:46-66; sunscreen/tests/features.rs:8-34; error collapse run.rs:78-82): every node's results are
  // watched in a status word owned by this run and read once at the end of every chunk
  hipStream_t s = (hipStream_t)stream;
  u32* status = g_throw_transparent ? (u32*)e->ev->scratch().acquire(256, s) : nullptr;
  if (g_throw_transparent && (!status || hipMemsetAsync(status, 0xFF, sizeof(u32), s) != hipSuccess)) {
    if (status) e->ev->scratch().release(status, s);
    return from_status(kOutOfMemory);
  }
  // The scheduled executor's table-driven kernels put (output, item) on grid z: a run takes at most kRunChunk input sets, and
  // a larger batch is a sequence of such runs over offset pointers (input sets are independent: the same bits) -- every kind of
  // argument, transform-domain plaintexts included, works at every batch size (beyond 32768 the node-by-node
  // executor used to take over, and it refuses kind 2).
  constexpr uint64_t kRunChunk = 32768;
  const size_t ct_words = e->ctx->ct_words(2);
  std::vector<u64*> outs(num_outputs);
  long hr = HIPBFV_S_OK;
  for (uint64_t off = 0; off < batch || (batch == 0 && off == 0); off += kRunChunk) {
    const uint64_t c = batch ? std::min<uint64_t>(kRunChunk, batch - off) : 0;  // an empty batch is the executor's to refuse
    for (uint64_t i = 0; i < num_inputs; i++) {
      ins[i] = ProgramInput{(int)input_kinds[i], (const u64*)input_ptrs[i], (size_t)input_strides[i]};
      if (!ins[i].ptr) continue;
      if (ins[i].kind == 0) ins[i].ptr += off * ct_words;
      else ins[i].ptr += off * ins[i].stride;  // stride 0: one shared plaintext
    }
    for (uint64_t k = 0; k < num_outputs; k++) outs[k] = outputs[k] ? (u64*)outputs[k] + off * ct_words : nullptr;
    keys.index = key_index ? key_index + off : nullptr;  // input set i of this chunk = input set off + i of the call
    keys.period = (size_t)c;
    int st;
    {
      WatchScope watch(status);
      st = p->prog.run(*e->ev, c, ins.data(), ins.size(), keys, outs.data(), num_outputs, s, &err);
    }
    u32 first_bad = 0xFFFFFFFFu;
    if (status) {
      const int st2 = e->ev->take_status(status, &first_bad, s);  // reads and resets the word
      if (st == kOk) st = st2;
    }
    if (st != kOk) {
      hr = from_status(st);
      if (!err.empty()) tls_error = err;
      break;
    }
    if (first_bad != 0xFFFFFFFFu) {
      char msg[128];
      // a merged launch numbers its ciphertexts member-major (member * batch + item): the item is what the caller knows
      snprintf(msg, sizeof(msg), "result ciphertext is transparent (input set %llu of the batch)", (unsigned long long)(off + first_bad % (c ? c : 1)));
      hr = fail(HIPBFV_COR_E_INVALIDOPERATION, msg);
      break;
    }
    if (!batch) break;
  }
  if (status) e->ev->scratch().release(status, s);
  return hr;
}

long hipbfv_Program_Run(void* h, void* evaluator, uint64_t batch, uint64_t num_inputs, const uint32_t* input_kinds,
                        const uint64_t* const* input_ptrs, const uint64_t* input_strides, void* relin_keys, void* galois_keys,
                        uint64_t num_outputs, uint64_t* const* outputs, void* stream) HIPBFV_BEGIN
  return program_run_impl(h, evaluator, batch, num_inputs, input_kinds, input_ptrs, input_strides, 1, &relin_keys, &galois_keys, nullptr, num_outputs,
                          outputs, stream);
HIPBFV_END

long hipbfv_Program_RunKeys(void* h, void* evaluator, uint64_t batch, uint64_t num_inputs, const uint32_t* input_kinds,
                            const uint64_t* const* input_ptrs, const uint64_t* input_strides, uint64_t num_key_sets, void* const* relin_keys,
                            void* const* galois_keys, const uint32_t* key_index, uint64_t num_outputs, uint64_t* const* outputs,
                            void* stream) HIPBFV_BEGIN
  if (!num_key_sets || !key_index) return HIPBFV_E_POINTER;
  return program_run_impl(h, evaluator, batch, num_inputs, input_kinds, input_ptrs, input_strides, num_key_sets, relin_keys, galois_keys, key_index,
                          num_outputs, outputs, stream);
HIPBFV_END

long hipbfv_batch_status(void* h, uint64_t* first_transparent_item, void* stream) HIPBFV_BEGIN
  EvalObj* e = as<EvalObj>(h, kMagicEval);
  if (!e) return HIPBFV_E_POINTER;
  u32 first_bad = 0xFFFFFFFFu;
  if (int st = e->ev->take_status(e->ev->batch_status(), &first_bad, (hipStream_t)stream)) return from_status(st);
  if (first_transparent_item) *first_transparent_item = first_bad == 0xFFFFFFFFu ? ~0ull : first_bad;
  if (first_bad != 0xFFFFFFFFu) {
    char msg[128];
    snprintf(msg, sizeof(msg), "result ciphertext is transparent (batch item %u)", first_bad);
    return fail(HIPBFV_COR_E_INVALIDOPERATION, msg);
  }
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_set_batch_transparent_check(void* h, bool enabled) HIPBFV_BEGIN
  EvalObj* e = as<EvalObj>(h, kMagicEval);
  if (!e) return HIPBFV_E_POINTER;
  e->batch_watch = enabled;
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_profile_enable(void* h, bool enabled) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  e->ev->profiler().collect();
  e->ev->profiler().enabled = enabled;
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_profile_reset(void* h) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  e->ev->profiler().reset();
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_profile_kernel_count(uint32_t* count) HIPBFV_BEGIN
  if (!count) return HIPBFV_E_POINTER;
  *count = kKernCount;
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_profile_read(void* h, uint32_t kernel_id, char* name, uint64_t name_capacity, double* total_ms, uint64_t* launches,
                         uint64_t* units) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  if (kernel_id >= (uint32_t)kKernCount) return fail(HIPBFV_E_INVALIDARG, "kernel id out of range");
  Profiler& p = e->ev->profiler();
  p.collect();
  if (name && name_capacity) {
    std::strncpy(name, kernel_name((int)kernel_id), name_capacity - 1);
    name[name_capacity - 1] = 0;
  }
  if (total_ms) *total_ms = p.total_ms[kernel_id];
  if (launches) *launches = p.launches[kernel_id];
  if (units) *units = p.units[kernel_id];
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_set_chunk_ops(void* h, uint64_t chunk) HIPBFV_BEGIN
  EVAL_OR_RETURN(h);
  e->ev->set_chunk_ops(chunk);
  return HIPBFV_S_OK;
HIPBFV_END

// ------------------------------------------------------------------ SecretKey / PublicKey (seal_fhe/src/key_generator.rs:200-430)
static long asym_create(uint32_t magic, void** out) {
  if (!out) return HIPBFV_E_POINTER;
  *out = new AsymKeyObj(magic);
  return HIPBFV_S_OK;
}
static long asym_copy(uint32_t magic, void* copy, void** out) {
  AsymKeyObj* k = as<AsymKeyObj>(copy, magic);
  if (!k || !out) return HIPBFV_E_POINTER;
  AsymKeyObj* n = new AsymKeyObj(magic);
  n->key = k->key;  // immutable once assigned: sharing is a deep copy as far as callers can tell
  *out = n;
  return HIPBFV_S_OK;
}
static long asym_destroy(uint32_t magic, void* h) {
  AsymKeyObj* k = as<AsymKeyObj>(h, magic);
  if (!k) return HIPBFV_E_POINTER;
  delete k;
  return HIPBFV_S_OK;
}
static long asym_assign(uint32_t magic, void* h, void* context, const uint64_t* host, size_t polys) {
  AsymKeyObj* k = as<AsymKeyObj>(h, magic);
  ContextObj* x = as<ContextObj>(context, kMagicContext);
  if (!k || !x || !host) return HIPBFV_E_POINTER;
  const Context& c = *x->ctx;
  const size_t words = polys * c.KK() * c.n();
  for (size_t p = 0; p < polys; p++)
    for (size_t i = 0; i < c.KK(); i++) {
      const u64 q = c.key_primes()[i];
      const uint64_t* row = host + (p * c.KK() + i) * c.n();
      for (size_t j = 0; j < c.n(); j++)
        if (row[j] >= q) return fail(HIPBFV_E_INVALIDARG, "key data is invalid for the encryption parameters");
    }
  auto buf = std::make_shared<KeyBuffer>();
  buf->ctx = x->ctx;
  buf->words = words;
  buf->dev = g_buffers.get(words);
  if (!buf->dev) return from_status(kOutOfMemory);
  if (hipMemcpy(buf->dev, host, words * sizeof(u64), hipMemcpyHostToDevice) != hipSuccess) return from_status(kHipError);
  k->key = buf;
  return HIPBFV_S_OK;
}
static long asym_to_host(AsymKeyObj* k, std::vector<u64>* host) {
  if (!k->key || !k->key->dev) return fail(HIPBFV_E_INVALIDARG, "key is empty");
  host->resize(k->key->words);
  if (hipMemcpy(host->data(), k->key->dev, host->size() * sizeof(u64), hipMemcpyDeviceToHost) != hipSuccess) return from_status(kHipError);
  return HIPBFV_S_OK;
}

long SecretKey_Create1(void** out) HIPBFV_BEGIN return asym_create(kMagicSecretKey, out); HIPBFV_END
long SecretKey_Create2(void* copy, void** out) HIPBFV_BEGIN return asym_copy(kMagicSecretKey, copy, out); HIPBFV_END
long SecretKey_Destroy(void* h) HIPBFV_BEGIN return asym_destroy(kMagicSecretKey, h); HIPBFV_END
long PublicKey_Create1(void** out) HIPBFV_BEGIN return asym_create(kMagicPublicKey, out); HIPBFV_END
long PublicKey_Create2(void* copy, void** out) HIPBFV_BEGIN return asym_copy(kMagicPublicKey, copy, out); HIPBFV_END
long PublicKey_Destroy(void* h) HIPBFV_BEGIN return asym_destroy(kMagicPublicKey, h); HIPBFV_END
long hipbfv_SecretKey_Assign(void* h, void* context, const uint64_t* host_data) HIPBFV_BEGIN return asym_assign(kMagicSecretKey, h, context, host_data, 1); HIPBFV_END
long hipbfv_PublicKey_Assign(void* h, void* context, const uint64_t* host_data) HIPBFV_BEGIN return asym_assign(kMagicPublicKey, h, context, host_data, 2); HIPBFV_END

static long asym_read(Magic m, void* h, uint64_t* host_out) {
  AsymKeyObj* k = as<AsymKeyObj>(h, m);
  if (!k || !host_out) return HIPBFV_E_POINTER;
  if (!k->key || !k->key->dev) return fail(HIPBFV_E_INVALIDARG, "key is empty");
  if (hipMemcpy(host_out, k->key->dev, k->key->words * sizeof(u64), hipMemcpyDeviceToHost) != hipSuccess) return from_status(kHipError);
  return HIPBFV_S_OK;
}
long hipbfv_SecretKey_Read(void* h, uint64_t* host_out) HIPBFV_BEGIN return asym_read(kMagicSecretKey, h, host_out); HIPBFV_END
long hipbfv_PublicKey_Read(void* h, uint64_t* host_out) HIPBFV_BEGIN return asym_read(kMagicPublicKey, h, host_out); HIPBFV_END
long hipbfv_KSwitchKeys_Read(void* h, uint64_t index, uint64_t* host_out) HIPBFV_BEGIN
  KeysObj* k = as<KeysObj>(h, kMagicKeys);
  if (!k || !host_out) return HIPBFV_E_POINTER;
  const u64* dev = k->find((u32)index);
  if (!dev) return fail(HIPBFV_E_INVALIDARG, "key not present");
  if (hipMemcpy(host_out, dev, k->ctx->key_words() * sizeof(u64), hipMemcpyDeviceToHost) != hipSuccess) return from_status(kHipError);
  return HIPBFV_S_OK;
HIPBFV_END
long hipbfv_KSwitchKeys_Has(void* h, uint64_t index, bool* present) HIPBFV_BEGIN
  KeysObj* k = as<KeysObj>(h, kMagicKeys);
  if (!k || !present) return HIPBFV_E_POINTER;
  *present = k->find((u32)index) != nullptr;
  return HIPBFV_S_OK;
HIPBFV_END

// SecretKey is serialised as a Plaintext whose parms_id is the key level's (seal_fhe/tests/data/secret_key.bin)
long SecretKey_SaveSize(void* h, uint8_t compr_mode, int64_t* result) HIPBFV_BEGIN
  AsymKeyObj* k = as<AsymKeyObj>(h, kMagicSecretKey);
  if (!k || !result) return HIPBFV_E_POINTER;
  const size_t raw = 16 + 32 + 16 + 16 + 8 + (k->key ? k->key->words : 0) * 8;
  *result = (int64_t)(compr_mode ? raw + raw / 128 + 512 : raw);
  return HIPBFV_S_OK;
HIPBFV_END
long SecretKey_Save(void* h, uint8_t* outptr, uint64_t size, uint8_t compr_mode, int64_t* out_bytes) HIPBFV_BEGIN
  AsymKeyObj* k = as<AsymKeyObj>(h, kMagicSecretKey);
  if (!k || !outptr || !out_bytes) return HIPBFV_E_POINTER;
  std::vector<u64> host;
  if (long hr = asym_to_host(k, &host)) return hr;
  uint8_t pid[32];
  key_level_parms_id(*k->key->ctx, pid);
  std::vector<uint8_t> buf;
  if (int rc = wire_pack_plaintext(pid, reinterpret_cast<const unsigned long long*>(host.data()), host.size(), compr_mode, &buf)) return from_wire(rc);
  if (size < buf.size()) return fail(HIPBFV_E_INVALIDARG, "buffer too small");
  std::memcpy(outptr, buf.data(), buf.size());
  *out_bytes = (int64_t)buf.size();
  return HIPBFV_S_OK;
HIPBFV_END
long SecretKey_Load(void* h, void* context, uint8_t* inptr, uint64_t size, int64_t* in_bytes) HIPBFV_BEGIN
  ContextObj* x = as<ContextObj>(context, kMagicContext);
  if (!as<AsymKeyObj>(h, kMagicSecretKey) || !x || !inptr || !in_bytes) return HIPBFV_E_POINTER;
  WirePlaintext pt;
  size_t used = 0;
  if (int rc = wire_unpack_plaintext(inptr, size, &pt, &used, max_plain_body(*x->ctx))) return from_wire(rc);
  uint8_t pid[32];
  key_level_parms_id(*x->ctx, pid);
  if (std::memcmp(pid, pt.parms_id, 32) != 0 || pt.coeffs.size() != (size_t)x->ctx->KK() * x->ctx->n())
    return fail(HIPBFV_E_INVALIDARG, "secret key data is invalid for the encryption parameters");
  long hr = asym_assign(kMagicSecretKey, h, context, reinterpret_cast<const uint64_t*>(pt.coeffs.data()), 1);
  if (hr == HIPBFV_S_OK) *in_bytes = (int64_t)used;
  return hr;
HIPBFV_END

// PublicKey is serialised as a size-2 NTT-form Ciphertext at the key level (seal_fhe/tests/data/public_key.bin)
long PublicKey_SaveSize(void* h, uint8_t compr_mode, int64_t* result) HIPBFV_BEGIN
  AsymKeyObj* k = as<AsymKeyObj>(h, kMagicPublicKey);
  if (!k || !result) return HIPBFV_E_POINTER;
  const size_t raw = 16 + 32 + 1 + 8 * 5 + 16 + 8 + (k->key ? k->key->words : 0) * 8;
  *result = (int64_t)(compr_mode ? raw + raw / 128 + 512 : raw);
  return HIPBFV_S_OK;
HIPBFV_END
long PublicKey_Save(void* h, uint8_t* outptr, uint64_t size, uint8_t compr_mode, int64_t* out_bytes) HIPBFV_BEGIN
  AsymKeyObj* k = as<AsymKeyObj>(h, kMagicPublicKey);
  if (!k || !outptr || !out_bytes) return HIPBFV_E_POINTER;
  std::vector<u64> host;
  if (long hr = asym_to_host(k, &host)) return hr;
  const Context& c = *k->key->ctx;
  uint8_t pid[32];
  key_level_parms_id(c, pid);
  std::vector<uint8_t> buf;
  if (int rc = wire_pack_ciphertext(pid, true, 2, c.n(), c.KK(), reinterpret_cast<const unsigned long long*>(host.data()), compr_mode, &buf))
    return from_wire(rc);
  if (size < buf.size()) return fail(HIPBFV_E_INVALIDARG, "buffer too small");
  std::memcpy(outptr, buf.data(), buf.size());
  *out_bytes = (int64_t)buf.size();
  return HIPBFV_S_OK;
HIPBFV_END
long PublicKey_Load(void* h, void* context, uint8_t* inptr, uint64_t size, int64_t* in_bytes) HIPBFV_BEGIN
  ContextObj* x = as<ContextObj>(context, kMagicContext);
  if (!as<AsymKeyObj>(h, kMagicPublicKey) || !x || !inptr || !in_bytes) return HIPBFV_E_POINTER;
  WireCiphertext ct;
  size_t used = 0;
  if (int rc = wire_unpack_ciphertext(inptr, size, &ct, &used, max_ct_body(*x->ctx))) return from_wire(rc);
  uint8_t pid[32];
  key_level_parms_id(*x->ctx, pid);
  if (std::memcmp(pid, ct.parms_id, 32) != 0 || !ct.is_ntt || ct.size != 2 || ct.n != x->ctx->n() || ct.k != x->ctx->KK())
    return fail(HIPBFV_E_INVALIDARG, "public key data is invalid for the encryption parameters");
  long hr = asym_assign(kMagicPublicKey, h, context, reinterpret_cast<const uint64_t*>(ct.data.data()), 2);
  if (hr == HIPBFV_S_OK) *in_bytes = (int64_t)used;
  return hr;
HIPBFV_END

// ------------------------------------------------------------------ BatchEncoder (seal_fhe/src/encoder.rs:50-215)
long BatchEncoder_Create(void* context, void** out) HIPBFV_BEGIN
  ContextObj* x = as<ContextObj>(context, kMagicContext);
  if (!x || !out) return HIPBFV_E_POINTER;
  if (!x->ctx->batching()) return fail(HIPBFV_E_INVALIDARG, "encryption parameters are not valid for batching");
  EncoderObj* e = new EncoderObj();
  e->ctx = x->ctx;
  e->ev.reset(new Evaluator(x->ctx.get()));
  *out = e;
  return HIPBFV_S_OK;
HIPBFV_END
long BatchEncoder_Destroy(void* h) HIPBFV_BEGIN
  EncoderObj* e = as<EncoderObj>(h, kMagicEncoder);
  if (!e) return HIPBFV_E_POINTER;
  delete e;
  return HIPBFV_S_OK;
HIPBFV_END
long BatchEncoder_GetSlotCount(void* h, uint64_t* count) HIPBFV_BEGIN
  EncoderObj* e = as<EncoderObj>(h, kMagicEncoder);
  if (!e || !count) return HIPBFV_E_POINTER;
  *count = e->ctx->n();
  return HIPBFV_S_OK;
HIPBFV_END
static long encode_common(void* h, uint64_t count, const uint64_t* values, void* plain, bool is_signed) {
  EncoderObj* e = as<EncoderObj>(h, kMagicEncoder);
  PlainObj* p = as<PlainObj>(plain, kMagicPlain);
  if (!e || !p || (count && !values)) return HIPBFV_E_POINTER;
  const size_t n = e->ctx->n();
  if (count > n) return fail(HIPBFV_E_INVALIDARG, "values has invalid size");
  std::vector<u64> slots(n, 0);
  std::copy(values, values + count, slots.begin());
  hipStream_t s = thread_stream();
  u64* dev = g_buffers.get(2 * n);
  if (!dev) return from_status(kOutOfMemory);
  long hr = HIPBFV_S_OK;
  u32 bad = 0;
  if (hipMemcpyAsync(dev, slots.data(), n * sizeof(u64), hipMemcpyHostToDevice, s) != hipSuccess)
    hr = from_status(kHipError);
  else if (int st = e->ev->batch_encode(dev, dev + n, 1, is_signed, &bad, s))
    hr = from_status(st);
  else if (bad)
    hr = fail(HIPBFV_E_INVALIDARG, "input value is larger than plain_modulus");
  else {
    p->coeffs.assign(n, 0);
    if (hipMemcpy(p->coeffs.data(), dev + n, n * sizeof(u64), hipMemcpyDeviceToHost) != hipSuccess) hr = from_status(kHipError);
  }
  g_buffers.put(dev, 2 * n);
  return hr;
}
long BatchEncoder_Encode1(void* h, uint64_t count, uint64_t* values, void* plain) HIPBFV_BEGIN return encode_common(h, count, values, plain, false); HIPBFV_END
long BatchEncoder_Encode2(void* h, uint64_t count, int64_t* values, void* plain) HIPBFV_BEGIN
  return encode_common(h, count, reinterpret_cast<const uint64_t*>(values), plain, true);
HIPBFV_END
static long decode_common(void* h, void* plain, uint64_t* count, uint64_t* values, bool is_signed) {
  EncoderObj* e = as<EncoderObj>(h, kMagicEncoder);
  PlainObj* p = as<PlainObj>(plain, kMagicPlain);
  if (!e || !p || !count || !values) return HIPBFV_E_POINTER;
  const size_t n = e->ctx->n();
  if (p->coeffs.size() > n) return fail(HIPBFV_E_INVALIDARG, "plain is not valid for encryption parameters");
  std::vector<u64> coeffs(n, 0);
  std::copy(p->coeffs.begin(), p->coeffs.end(), coeffs.begin());
  hipStream_t s = thread_stream();
  u64* dev = g_buffers.get(2 * n);
  if (!dev) return from_status(kOutOfMemory);
  long hr = HIPBFV_S_OK;
  if (hipMemcpyAsync(dev, coeffs.data(), n * sizeof(u64), hipMemcpyHostToDevice, s) != hipSuccess)
    hr = from_status(kHipError);
  else if (int st = e->ev->batch_decode(dev, dev + n, 1, is_signed, s))
    hr = from_status(st);
  else if (hipMemcpyAsync(values, dev + n, n * sizeof(u64), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
    hr = from_status(kHipError);
  else
    *count = n;
  g_buffers.put(dev, 2 * n);
  return hr;
}
long BatchEncoder_Decode1(void* h, void* plain, uint64_t* count, uint64_t* values, void* pool) HIPBFV_BEGIN
  (void)pool;
  return decode_common(h, plain, count, values, false);
HIPBFV_END
long BatchEncoder_Decode2(void* h, void* plain, uint64_t* count, int64_t* values, void* pool) HIPBFV_BEGIN
  (void)pool;
  return decode_common(h, plain, count, reinterpret_cast<uint64_t*>(values), true);
HIPBFV_END

// ------------------------------------------------------------------ Decryptor (seal_fhe/src/encryptor_decryptor.rs:596-690)
long Decryptor_Create(void* context, void* secret_key, void** out) HIPBFV_BEGIN
  ContextObj* x = as<ContextObj>(context, kMagicContext);
  AsymKeyObj* k = as<AsymKeyObj>(secret_key, kMagicSecretKey);
  if (!x || !k || !out) return HIPBFV_E_POINTER;
  if (!k->key || k->key->ctx.get() != x->ctx.get()) return fail(HIPBFV_E_INVALIDARG, "secret key is not valid for encryption parameters");
  DecryptorObj* d = new DecryptorObj();
  d->ctx = x->ctx;
  d->core.ctx = x->ctx;
  d->core.ev.reset(new Evaluator(x->ctx.get()));
  d->sk = k->key;
  *out = d;
  return HIPBFV_S_OK;
HIPBFV_END
long Decryptor_Destroy(void* h) HIPBFV_BEGIN
  DecryptorObj* d = as<DecryptorObj>(h, kMagicDecryptor);
  if (!d) return HIPBFV_E_POINTER;
  delete d;
  return HIPBFV_S_OK;
HIPBFV_END
long Decryptor_Decrypt(void* h, void* encrypted, void* destination) HIPBFV_BEGIN
  DecryptorObj* d = as<DecryptorObj>(h, kMagicDecryptor);
  CipherObj* c = as<CipherObj>(encrypted, kMagicCipher);
  PlainObj* p = as<PlainObj>(destination, kMagicPlain);
  if (!d || !c || !p) return HIPBFV_E_POINTER;
  EvalObj* le = c->ctx ? level_eval(&d->core, c->ctx) : nullptr;
  if (!le || !c->dev || c->size < 2) return fail(HIPBFV_E_INVALIDARG, "encrypted is not valid for encryption parameters");
  const size_t n = d->ctx->n();
  hipStream_t s = thread_stream();
  u64* dev = g_buffers.get(n);
  if (!dev) return from_status(kOutOfMemory);
  long hr = HIPBFV_S_OK;
  std::vector<u64> host(n);
  // the secret key's residues for a lower level are a prefix of its rows (the data primes come first)
  if (int st = le->ev->decrypt(c->dev, c->size, d->sk->dev, dev, 1, s))
    hr = from_status(st);
  else if (hipMemcpyAsync(host.data(), dev, n * sizeof(u64), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
    hr = from_status(kHipError);
  g_buffers.put(dev, n);
  if (hr != HIPBFV_S_OK) return hr;
  // SEAL resizes the plaintext to its significant coefficient count (at least one coefficient)
  size_t len = n;
  while (len > 1 && host[len - 1] == 0) len--;
  host.resize(len);
  p->coeffs.swap(host);
  return HIPBFV_S_OK;
HIPBFV_END

// Decryptor::invariant_noise_budget (encryptor_decryptor.rs:640-660): bits(q) - bits(|t * phase mod q| centred, max over
// coefficients) - 1, floored at 0.  The phase comes from the device; the multi-word arithmetic (one CRT composition
// per coefficient) runs on the host -- a diagnostic, not a hot path.
namespace {
struct Big {
  std::vector<u64> w;  // little-endian, fixed length
  explicit Big(size_t len) : w(len, 0) {}
  void add_mul(const Big& a, u64 m) {  // *this += a * m  (no overflow by construction: one spare word)
    u64 carry = 0;
    for (size_t i = 0; i < w.size(); i++) {
      const unsigned __int128 p = (unsigned __int128)(i < a.w.size() ? a.w[i] : 0) * m + w[i] + carry;
      w[i] = (u64)p;
      carry = (u64)(p >> 64);
    }
  }
  int cmp(const Big& o) const {
    for (size_t i = w.size(); i-- > 0;) {
      if (w[i] != o.w[i]) return w[i] < o.w[i] ? -1 : 1;
    }
    return 0;
  }
  void sub(const Big& o) {
    u64 borrow = 0;
    for (size_t i = 0; i < w.size(); i++) {
      const unsigned __int128 d = (unsigned __int128)w[i] - o.w[i] - borrow;
      w[i] = (u64)d;
      borrow = (u64)(d >> 64) & 1;
    }
  }
  int bits() const {
    for (size_t i = w.size(); i-- > 0;)
      if (w[i]) return (int)(64 * i) + 64 - __builtin_clzll(w[i]);
    return 0;
  }
};
static u64 mulmod64(u64 a, u64 b, u64 q) { return (u64)((unsigned __int128)a * b % q); }
static u64 invmod64(u64 a, u64 q) {  // q prime
  u64 r = 1, e = q - 2;
  a %= q;
  while (e) {
    if (e & 1) r = mulmod64(r, a, q);
    a = mulmod64(a, a, q);
    e >>= 1;
  }
  return r;
}
}  // namespace

// max_x |[t * ct(s)]_q|_x (centred) and q, as big integers: the quantity behind both noise measures
static long invariant_noise_norm(void* h, void* encrypted, Big* worst_out, Big* q_out) {
  DecryptorObj* d = as<DecryptorObj>(h, kMagicDecryptor);
  CipherObj* c = as<CipherObj>(encrypted, kMagicCipher);
  if (!d || !c) return HIPBFV_E_POINTER;
  EvalObj* le = c->ctx ? level_eval(&d->core, c->ctx) : nullptr;
  if (!le || !c->dev || c->size < 2) return fail(HIPBFV_E_INVALIDARG, "encrypted is not valid for encryption parameters");
  const Context& cx = *le->ctx;
  const size_t n = cx.n(), K = cx.K();
  hipStream_t s = thread_stream();
  u64* dev = g_buffers.get(K * n);
  if (!dev) return from_status(kOutOfMemory);
  std::vector<u64> ph(K * n);
  long hr = HIPBFV_S_OK;
  if (int st = le->ev->phase(c->dev, c->size, d->sk->dev, dev, 1, s))
    hr = from_status(st);
  else if (hipMemcpyAsync(ph.data(), dev, ph.size() * sizeof(u64), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
    hr = from_status(kHipError);
  g_buffers.put(dev, K * n);
  if (hr != HIPBFV_S_OK) return hr;
  const std::vector<u64>& q = cx.key_primes();
  const size_t len = K + 1;
  Big Q(len);
  Q.w[0] = 1;
  for (size_t i = 0; i < K; i++) {
    Big t2(len);
    t2.add_mul(Q, q[i]);
    Q = t2;
  }
  Big half = Q;  // floor(q / 2)
  for (size_t i = 0; i < len; i++) half.w[i] = (Q.w[i] >> 1) | (i + 1 < len ? Q.w[i + 1] << 63 : 0);
  std::vector<Big> punct(K, Big(len));  // q / q_i
  std::vector<u64> scale(K);            // t * (q/q_i)^{-1} mod q_i
  for (size_t i = 0; i < K; i++) {
    punct[i].w[0] = 1;
    u64 pm = 1;
    for (size_t j = 0; j < K; j++) {
      if (j == i) continue;
      Big t2(len);
      t2.add_mul(punct[i], q[j]);
      punct[i] = t2;
      pm = mulmod64(pm, q[j] % q[i], q[i]);
    }
    scale[i] = mulmod64(cx.t() % q[i], invmod64(pm, q[i]), q[i]);
  }
  Big worst(len);
  for (size_t x = 0; x < n; x++) {
    Big v(len);
    for (size_t i = 0; i < K; i++) v.add_mul(punct[i], mulmod64(ph[i * n + x], scale[i], q[i]));
    while (v.cmp(Q) >= 0) v.sub(Q);
    if (v.cmp(half) > 0) {
      Big r = Q;
      r.sub(v);
      v = r;
    }
    if (v.cmp(worst) > 0) worst = v;
  }
  *worst_out = worst;
  *q_out = Q;
  return HIPBFV_S_OK;
}
long Decryptor_InvariantNoiseBudget(void* h, void* encrypted, int* budget) HIPBFV_BEGIN
  if (!budget) return HIPBFV_E_POINTER;
  Big worst(1), Q(1);
  if (long hr = invariant_noise_norm(h, encrypted, &worst, &Q)) return hr;
  *budget = std::max(0, Q.bits() - worst.bits() - 1);
  return HIPBFV_S_OK;
HIPBFV_END
// the fork's f64 variant (encryptor_decryptor.rs:660-683): the infinity norm of the invariant noise polynomial,
// |[t * ct(s)]_q| / q; decryption is correct while it stays below 1/2
long Decryptor_InvariantNoise(void* h, void* encrypted, double* invariant_noise) HIPBFV_BEGIN
  if (!invariant_noise) return HIPBFV_E_POINTER;
  Big worst(1), Q(1);
  if (long hr = invariant_noise_norm(h, encrypted, &worst, &Q)) return hr;
  auto to_ld = [](const Big& b) {
    long double v = 0;
    for (size_t i = b.w.size(); i-- > 0;) v = v * 18446744073709551616.0L + (long double)b.w[i];
    return v;
  };
  *invariant_noise = (double)(to_ld(worst) / to_ld(Q));
  return HIPBFV_S_OK;
HIPBFV_END

// ------------------------------------------------------------------ Encryptor, public-key mode (encryptor_decryptor.rs:140-260)
long Encryptor_Create(void* context, void* public_key, void* secret_key, void** out) HIPBFV_BEGIN
  ContextObj* x = as<ContextObj>(context, kMagicContext);
  if (!x || !out) return HIPBFV_E_POINTER;
  // with_public_key / with_secret_key / with_public_and_secret_key (encryptor_decryptor.rs:140-200): either may be null
  AsymKeyObj* pk = public_key ? as<AsymKeyObj>(public_key, kMagicPublicKey) : nullptr;
  AsymKeyObj* sk = secret_key ? as<AsymKeyObj>(secret_key, kMagicSecretKey) : nullptr;
  if ((public_key && !pk) || (secret_key && !sk)) return HIPBFV_E_POINTER;
  if (!pk && !sk) return fail(HIPBFV_E_INVALIDARG, "a public key or a secret key is required");
  if (pk && (!pk->key || pk->key->ctx.get() != x->ctx.get())) return fail(HIPBFV_E_INVALIDARG, "public key is not valid for encryption parameters");
  if (sk && (!sk->key || sk->key->ctx.get() != x->ctx.get())) return fail(HIPBFV_E_INVALIDARG, "secret key is not valid for encryption parameters");
  EncryptorObj* e = new EncryptorObj();
  e->ctx = x->ctx;
  e->ev.reset(new Evaluator(x->ctx.get()));
  if (pk) e->pk = pk->key;
  if (sk) e->sk = sk->key;
  // fresh 512-bit seed per Encryptor from the OS (SEAL seeds its PRNG factory from the OS the same way)
  if (!os_seed(&e->seed)) {
    delete e;
    return fail(HIPBFV_E_UNEXPECTED, "no entropy available from the operating system (getrandom failed)");
  }
  *out = e;
  return HIPBFV_S_OK;
HIPBFV_END
long Encryptor_Destroy(void* h) HIPBFV_BEGIN
  EncryptorObj* e = as<EncryptorObj>(h, kMagicEncryptor);
  if (!e) return HIPBFV_E_POINTER;
  delete e;
  return HIPBFV_S_OK;
HIPBFV_END
long hipbfv_Encryptor_SetSeed(void* h, uint64_t seed) HIPBFV_BEGIN
  EncryptorObj* e = as<EncryptorObj>(h, kMagicEncryptor);
  if (!e) return HIPBFV_E_POINTER;
  std::lock_guard<std::mutex> g(e->mu);
  e->seed = rng_seed_from_u64_for_tests(seed);  // TEST ONLY: 64 bits of entropy (reproducible ciphertexts for parity tests)
  e->next_op = 0;
  return HIPBFV_S_OK;
HIPBFV_END
long Encryptor_Encrypt(void* h, void* plaintext, void* destination, void* pool) HIPBFV_BEGIN
  (void)pool;
  EncryptorObj* e = as<EncryptorObj>(h, kMagicEncryptor);
  PlainObj* p = as<PlainObj>(plaintext, kMagicPlain);
  CipherObj* c = as<CipherObj>(destination, kMagicCipher);
  if (!e || !p || !c) return HIPBFV_E_POINTER;
  if (!e->pk) return fail(HIPBFV_COR_E_INVALIDOPERATION, "public key is not set");
  const size_t n = e->ctx->n(), words = e->ctx->ct_words(2);
  if (p->coeffs.size() > n) return fail(HIPBFV_E_INVALIDARG, "plain is not valid for encryption parameters");
  for (u64 v : p->coeffs)
    if (v >= e->ctx->t()) return fail(HIPBFV_E_INVALIDARG, "plain is not valid for encryption parameters");
  u64 op;
  RngSeed seed;
  {
    std::lock_guard<std::mutex> g(e->mu);
    op = e->next_op++;
    seed = e->seed;
  }
  std::vector<u64> coeffs(n, 0);
  std::copy(p->coeffs.begin(), p->coeffs.end(), coeffs.begin());
  hipStream_t s = thread_stream();
  u64* pl = g_buffers.get(n);
  u64* out = g_buffers.get(words);
  if (!pl || !out) {
    g_buffers.put(pl, n);
    g_buffers.put(out, words);
    return from_status(kOutOfMemory);
  }
  long hr = HIPBFV_S_OK;
  if (hipMemcpyAsync(pl, coeffs.data(), n * sizeof(u64), hipMemcpyHostToDevice, s) != hipSuccess)
    hr = from_status(kHipError);
  else if (int st = e->ev->encrypt(pl, 0, e->pk->dev, seed, op, out, 1, s))
    hr = from_status(st);
  else
    hr = sync_stream(s);
  g_buffers.put(pl, n);
  if (hr != HIPBFV_S_OK) {
    g_buffers.put(out, words);
    return hr;
  }
  c->adopt(e->ctx, 2, out, words);
  return HIPBFV_S_OK;
HIPBFV_END

// ------------------------------------------------------------------ batched device-pointer forms
static Evaluator* eval_of(void* evaluator) {
  EvalObj* e = as<EvalObj>(evaluator, kMagicEval);
  return e ? e->ev.get() : nullptr;
}
long hipbfv_batch_encode(void* evaluator, const uint64_t* values, uint64_t* plain, uint64_t count, int is_signed, void* stream) HIPBFV_BEGIN
  Evaluator* ev = eval_of(evaluator);
  if (!ev || !values || !plain) return HIPBFV_E_POINTER;
  u32 bad = 0;
  if (int st = ev->batch_encode((const u64*)values, (u64*)plain, count, is_signed != 0, &bad, (hipStream_t)stream)) return from_status(st);
  if (bad) return fail(HIPBFV_E_INVALIDARG, "input value is larger than plain_modulus");
  return HIPBFV_S_OK;
HIPBFV_END
long hipbfv_batch_decode(void* evaluator, const uint64_t* plain, uint64_t* values, uint64_t count, int is_signed, void* stream) HIPBFV_BEGIN
  Evaluator* ev = eval_of(evaluator);
  if (!ev || !values || !plain) return HIPBFV_E_POINTER;
  return from_status(ev->batch_decode((const u64*)plain, (u64*)values, count, is_signed != 0, (hipStream_t)stream));
HIPBFV_END
long hipbfv_batch_decrypt(void* evaluator, const uint64_t* ct, uint32_t size, void* secret_key, uint64_t* plain, uint64_t count, void* stream) HIPBFV_BEGIN
  EvalObj* e = as<EvalObj>(evaluator, kMagicEval);
  AsymKeyObj* k = as<AsymKeyObj>(secret_key, kMagicSecretKey);
  if (!e || !k || !ct || !plain) return HIPBFV_E_POINTER;
  if (!k->key || k->key->ctx.get() != e->ctx.get()) return fail(HIPBFV_E_INVALIDARG, "secret key is not valid for encryption parameters");
  return from_status(e->ev->decrypt((const u64*)ct, size, k->key->dev, (u64*)plain, count, (hipStream_t)stream));
HIPBFV_END
long hipbfv_batch_encrypt(void* evaluator, const uint64_t* plain, uint64_t plain_stride, void* public_key, uint64_t seed, uint64_t first_op,
                          uint64_t* ct, uint64_t count, void* stream) HIPBFV_BEGIN
  EvalObj* e = as<EvalObj>(evaluator, kMagicEval);
  AsymKeyObj* k = as<AsymKeyObj>(public_key, kMagicPublicKey);
  if (!e || !k || !ct || !plain) return HIPBFV_E_POINTER;
  if (!k->key || k->key->ctx.get() != e->ctx.get()) return fail(HIPBFV_E_INVALIDARG, "public key is not valid for encryption parameters");
  // TEST ONLY: a 64-bit seed (reproducible batches for the parity tests); production callers use hipbfv_batch_encrypt_seeded
  return from_status(e->ev->encrypt((const u64*)plain, plain_stride, k->key->dev, rng_seed_from_u64_for_tests(seed), first_op, (u64*)ct, count, (hipStream_t)stream));
HIPBFV_END
long hipbfv_batch_encrypt_seeded(void* evaluator, const uint64_t* plain, uint64_t plain_stride, void* public_key, const uint8_t* seed64, uint64_t first_op,
                                 uint64_t* ct, uint64_t count, void* stream) HIPBFV_BEGIN
  EvalObj* e = as<EvalObj>(evaluator, kMagicEval);
  AsymKeyObj* k = as<AsymKeyObj>(public_key, kMagicPublicKey);
  if (!e || !k || !ct || !plain) return HIPBFV_E_POINTER;
  if (!k->key || k->key->ctx.get() != e->ctx.get()) return fail(HIPBFV_E_INVALIDARG, "public key is not valid for encryption parameters");
  RngSeed seed;
  if (seed64)
    seed = rng_seed_from_512(seed64);
  else if (!os_seed(&seed))
    return fail(HIPBFV_E_UNEXPECTED, "no entropy available from the operating system (getrandom failed)");
  return from_status(e->ev->encrypt((const u64*)plain, plain_stride, k->key->dev, seed, first_op, (u64*)ct, count, (hipStream_t)stream));
HIPBFV_END

// ------------------------------------------------------------------ plaintext-matrix x ciphertext-vector (PIR, examples/pir)
// A transformed plaintext is only ever consumed by a ciphertext-plaintext product, and SEAL refuses the product with an all-zero
// plaintext (transparent result; sunscreen/tests/features.rs:8-34).  The consumers (dot_plain_ntt, Program_Run's kind-2
// arguments) no longer see coefficients, so the PRODUCER records the first all-zero plaintext in the evaluator's status word,
// like every other hipbfv_batch_* operation records its transparent results: hipbfv_batch_status reports it.
long hipbfv_batch_plain_to_ntt(void* evaluator, const uint64_t* plain, uint64_t plain_stride, uint64_t* pntt, uint64_t count, void* stream) HIPBFV_BEGIN
  EVAL_OR_RETURN(evaluator);
  if (!plain || !pntt) return HIPBFV_E_POINTER;
  return from_status(e->ev->plain_to_ntt((const u64*)plain, plain_stride, (u64*)pntt, count, (hipStream_t)stream, 1));
HIPBFV_END
long hipbfv_batch_ct_to_ntt(void* evaluator, const uint64_t* ct, uint64_t size, uint64_t* ctn, uint64_t count, void* stream) HIPBFV_BEGIN
  Evaluator* ev = eval_of(evaluator);
  if (!ev || !ct || !ctn) return HIPBFV_E_POINTER;
  if (size < 1) return fail(HIPBFV_E_INVALIDARG, "invalid ciphertext size");
  return from_status(ev->ct_to_ntt((const u64*)ct, (u32)size, (u64*)ctn, count, (hipStream_t)stream));
HIPBFV_END
long hipbfv_batch_dot_plain_ntt(void* evaluator, const uint64_t* ctn, uint64_t cols, const uint64_t* pntt, uint64_t rows, uint64_t* out,
                                void* stream) HIPBFV_BEGIN
  Evaluator* ev = eval_of(evaluator);
  if (!ev || !ctn || !pntt || !out) return HIPBFV_E_POINTER;
  if (cols > 0xFFFFFFFFull || rows > 0xFFFFFFFFull) return fail(HIPBFV_E_INVALIDARG, "matrix too large");
  return from_status(ev->dot_plain_ntt((const u64*)ctn, (u32)cols, (const u64*)pntt, (u32)rows, (u64*)out, (hipStream_t)stream));
HIPBFV_END

// ------------------------------------------------------------------ modulus switching (evaluator_base.rs: mod_switch_to_next)
long Evaluator_ModSwitchToNext1(void* h, void* encrypted, void* destination, void* pool) HIPBFV_BEGIN
  (void)pool;
  EvalObj* e = as<EvalObj>(h, kMagicEval);
  CipherObj *x = as<CipherObj>(encrypted, kMagicCipher), *d = as<CipherObj>(destination, kMagicCipher);
  if (!e || !x || !d) return HIPBFV_E_POINTER;
  if (EvalObj* le = level_eval(e, x->ctx)) e = le;
  if (!same_context(x, e)) return fail(HIPBFV_E_INVALIDARG, "encrypted is not valid for encryption parameters");
  std::string err;
  std::shared_ptr<Context> next = e->ctx->next_level(&err);
  if (!next) return fail(HIPBFV_E_INVALIDARG, err.empty() ? "end of modulus switching chain reached" : err.c_str());
  hipStream_t s = thread_stream();
  const size_t words = next->ct_words(x->size);
  u64* buf = g_buffers.get(words);
  if (!buf) return from_status(kOutOfMemory);
  int st = e->ev->mod_switch_next(x->dev, x->size, buf, 1, s);
  if (st) {
    g_buffers.put(buf, words);
    return from_status(st);
  }
  EvalObj* ne = level_eval(e, next);  // e may itself be a lower level: resolve from the level's own chain
  if (!ne) {
    g_buffers.put(buf, words);
    return from_status(kHipError);
  }
  return finish_result(ne, d, x->size, buf, words, s, true);
HIPBFV_END

long Evaluator_ModSwitchToNext2(void* h, void* plain, void* destination) HIPBFV_BEGIN
  // SEAL mod-switches only NTT-form plaintexts (CKKS / pre-transformed); BFV plaintexts of this path are never in NTT form
  if (!as<EvalObj>(h, kMagicEval) || !as<PlainObj>(plain, kMagicPlain) || !as<PlainObj>(destination, kMagicPlain)) return HIPBFV_E_POINTER;
  return fail(HIPBFV_E_INVALIDARG, "plain is not in NTT form");
HIPBFV_END

long hipbfv_Context_NextLevel(void* context, void** next) HIPBFV_BEGIN
  ContextObj* x = as<ContextObj>(context, kMagicContext);
  if (!x || !next) return HIPBFV_E_POINTER;
  std::string err;
  std::shared_ptr<Context> n = x->ctx->next_level(&err);
  if (!n) return fail(HIPBFV_E_INVALIDARG, err.empty() ? "end of modulus switching chain reached" : err.c_str());
  ContextObj* o = new ContextObj();
  o->ctx = n;
  *next = o;
  return HIPBFV_S_OK;
HIPBFV_END

long hipbfv_batch_mod_switch(void* evaluator, const uint64_t* ct, uint64_t size, uint64_t* out, uint64_t count, void* stream) HIPBFV_BEGIN
  EVAL_OR_RETURN(evaluator);
  Evaluator* ev = e->ev.get();
  if (!ct || !out) return HIPBFV_E_POINTER;
  if (size < 1) return fail(HIPBFV_E_INVALIDARG, "invalid ciphertext size");
  return from_status(ev->mod_switch_next((const u64*)ct, (u32)size, (u64*)out, count, (hipStream_t)stream));
HIPBFV_END

// ------------------------------------------------------------------ PolynomialArray + encryption components (fork-only API)
static long polyarray_fill(PolyArrayObj* a, const std::shared_ptr<Context>& ctx, u32 polys, u32 kc) {
  if (a->reserved) return fail(HIPBFV_COR_E_INVALIDOPERATION, "polynomial array already holds data");
  a->ctx = ctx;
  a->polys = polys;
  a->kc = kc;
  a->words = (size_t)polys * kc * ctx->n();
  a->dev = a->words ? g_buffers.get(a->words) : nullptr;
  if (a->words && !a->dev) return from_status(kOutOfMemory);
  a->rns = true;
  a->reserved = true;
  return HIPBFV_S_OK;
}
long PolynomialArray_Create(void* pool, void** out) HIPBFV_BEGIN
  (void)pool;
  if (!out) return HIPBFV_E_POINTER;
  *out = new PolyArrayObj();
  return HIPBFV_S_OK;
HIPBFV_END
long PolynomialArray_Destroy(void* h) HIPBFV_BEGIN
  PolyArrayObj* a = as<PolyArrayObj>(h, kMagicPolyArray);
  if (!a) return HIPBFV_E_POINTER;
  delete a;
  return HIPBFV_S_OK;
HIPBFV_END
long PolynomialArray_CreateFromCiphertext(void* pool, void* context, void* ciphertext, void** out) HIPBFV_BEGIN
  (void)pool;
  ContextObj* x = as<ContextObj>(context, kMagicContext);
  CipherObj* c = as<CipherObj>(ciphertext, kMagicCipher);
  if (!x || !c || !out) return HIPBFV_E_POINTER;
  std::shared_ptr<Context> ctx = c->ctx ? c->ctx : x->ctx;
  if (ctx->n() != x->ctx->n()) return fail(HIPBFV_E_INVALIDARG, "ciphertext is not valid for encryption parameters");
  std::unique_ptr<PolyArrayObj> a(new PolyArrayObj());
  const u32 polys = c->dev ? c->size : 0;
  if (long hr = polyarray_fill(a.get(), ctx, polys, ctx->K())) return hr;
  if (a->words) {
    hipStream_t s = thread_stream();
    if (hipMemcpyAsync(a->dev, c->dev, a->words * sizeof(u64), hipMemcpyDeviceToDevice, s) != hipSuccess) return from_status(kHipError);
    if (long hr = sync_stream(s)) return hr;
  }
  *out = a.release();
  return HIPBFV_S_OK;
HIPBFV_END
static long polyarray_from_key(void* context, AsymKeyObj* k, u32 polys, void** out) {
  ContextObj* x = as<ContextObj>(context, kMagicContext);
  if (!x || !k || !out) return HIPBFV_E_POINTER;
  if (!k->key || k->key->ctx.get() != x->ctx.get()) return fail(HIPBFV_E_INVALIDARG, "key is not valid for encryption parameters");
  std::unique_ptr<PolyArrayObj> a(new PolyArrayObj());
  if (long hr = polyarray_fill(a.get(), x->ctx, polys, x->ctx->K())) return hr;
  Evaluator ev(x->ctx.get());
  hipStream_t s = thread_stream();
  // keys live at the key level in NTT form; the array holds the data-level residues in coefficient form
  if (int st = ev.key_to_coeff(k->key->dev, polys, a->dev, s)) return from_status(st);
  if (long hr = sync_stream(s)) return hr;
  *out = a.release();
  return HIPBFV_S_OK;
}
long PolynomialArray_CreateFromPublicKey(void* pool, void* context, void* public_key, void** out) HIPBFV_BEGIN
  (void)pool;
  return polyarray_from_key(context, as<AsymKeyObj>(public_key, kMagicPublicKey), 2, out);
HIPBFV_END
long PolynomialArray_CreateFromSecretKey(void* pool, void* context, void* secret_key, void** out) HIPBFV_BEGIN
  (void)pool;
  return polyarray_from_key(context, as<AsymKeyObj>(secret_key, kMagicSecretKey), 1, out);
HIPBFV_END
long PolynomialArray_Copy(void* h, void** out) HIPBFV_BEGIN
  PolyArrayObj* a = as<PolyArrayObj>(h, kMagicPolyArray);
  if (!a || !out) return HIPBFV_E_POINTER;
  std::unique_ptr<PolyArrayObj> b(new PolyArrayObj());
  if (a->reserved) {
    if (long hr = polyarray_fill(b.get(), a->ctx, a->polys, a->kc)) return hr;
    b->rns = a->rns;
    if (a->words) {
      hipStream_t s = thread_stream();
      if (hipMemcpyAsync(b->dev, a->dev, a->words * sizeof(u64), hipMemcpyDeviceToDevice, s) != hipSuccess) return from_status(kHipError);
      if (long hr = sync_stream(s)) return hr;
    }
  }
  *out = b.release();
  return HIPBFV_S_OK;
HIPBFV_END
long PolynomialArray_IsReserved(void* h, bool* result) HIPBFV_BEGIN
  PolyArrayObj* a = as<PolyArrayObj>(h, kMagicPolyArray);
  if (!a || !result) return HIPBFV_E_POINTER;
  *result = a->reserved;
  return HIPBFV_S_OK;
HIPBFV_END
long PolynomialArray_IsRns(void* h, bool* result) HIPBFV_BEGIN
  PolyArrayObj* a = as<PolyArrayObj>(h, kMagicPolyArray);
  if (!a || !result) return HIPBFV_E_POINTER;
  *result = a->rns;
  return HIPBFV_S_OK;
HIPBFV_END
long PolynomialArray_PolySize(void* h, uint64_t* result) HIPBFV_BEGIN
  PolyArrayObj* a = as<PolyArrayObj>(h, kMagicPolyArray);
  if (!a || !result) return HIPBFV_E_POINTER;
  *result = a->polys;
  return HIPBFV_S_OK;
HIPBFV_END
long PolynomialArray_PolyModulusDegree(void* h, uint64_t* result) HIPBFV_BEGIN
  PolyArrayObj* a = as<PolyArrayObj>(h, kMagicPolyArray);
  if (!a || !result) return HIPBFV_E_POINTER;
  *result = a->ctx ? a->ctx->n() : 0;
  return HIPBFV_S_OK;
HIPBFV_END
long PolynomialArray_CoeffModulusSize(void* h, uint64_t* result) HIPBFV_BEGIN
  PolyArrayObj* a = as<PolyArrayObj>(h, kMagicPolyArray);
  if (!a || !result) return HIPBFV_E_POINTER;
  *result = a->kc;
  return HIPBFV_S_OK;
HIPBFV_END
long PolynomialArray_ExportSize(void* h, uint64_t* result) HIPBFV_BEGIN
  PolyArrayObj* a = as<PolyArrayObj>(h, kMagicPolyArray);
  if (!a || !result) return HIPBFV_E_POINTER;
  *result = a->words;
  return HIPBFV_S_OK;
HIPBFV_END
long PolynomialArray_PerformExport(void* h, uint64_t* data) HIPBFV_BEGIN
  PolyArrayObj* a = as<PolyArrayObj>(h, kMagicPolyArray);
  if (!a || (!data && a->words)) return HIPBFV_E_POINTER;
  if (a->words && hipMemcpy(data, a->dev, a->words * sizeof(u64), hipMemcpyDeviceToHost) != hipSuccess) return from_status(kHipError);
  return HIPBFV_S_OK;
HIPBFV_END
// CRT constants of the first kc primes: inv_punct[kc] | punct[kc][kc] | q[kc]
static std::vector<u64> crt_constants(const Context& cx, u32 kc) {
  std::vector<u64> c((size_t)kc + (size_t)kc * kc + kc, 0);
  const DevCtx& h = cx.host();
  for (u32 i = 0; i < kc; i++) {
    Big punct(kc + 1);
    punct.w[0] = 1;
    u64 pm = 1;
    for (u32 j = 0; j < kc; j++) {
      if (j == i) continue;
      Big next(kc + 1);
      next.add_mul(punct, h.mod[j].q);
      punct = next;
      pm = mulmod64(pm, h.mod[j].q % h.mod[i].q, h.mod[i].q);
    }
    c[i] = invmod64(pm, h.mod[i].q);
    for (u32 l = 0; l < kc; l++) c[kc + (size_t)i * kc + l] = punct.w[l];
    if (i == 0) {
      Big q(kc + 1);
      q.add_mul(punct, h.mod[0].q);
      for (u32 l = 0; l < kc; l++) c[kc + (size_t)kc * kc + l] = q.w[l];
    }
  }
  return c;
}
static long polyarray_convert(PolyArrayObj* a, bool to_rns) {
  if (!a->reserved || a->rns == to_rns || !a->words) {
    if (a->reserved) a->rns = to_rns;
    return HIPBFV_S_OK;
  }
  u64* out = g_buffers.get(a->words);
  if (!out) return from_status(kOutOfMemory);
  hipStream_t s = thread_stream();
  long hr = HIPBFV_S_OK;
  const u32 n = a->ctx->n();
  if (to_rns) {
    if (launch_crt_decompose(a->ctx->dev(), n, a->kc, a->dev, out, a->polys, s) != hipSuccess) hr = from_status(kHipError);
  } else {
    const std::vector<u64> consts = crt_constants(*a->ctx, a->kc);
    u64* dc = g_buffers.get(consts.size());
    if (!dc) {
      g_buffers.put(out, a->words);
      return from_status(kOutOfMemory);
    }
    if (hipMemcpyAsync(dc, consts.data(), consts.size() * sizeof(u64), hipMemcpyHostToDevice, s) != hipSuccess ||
        launch_crt_compose(a->ctx->dev(), n, a->kc, dc, a->dev, out, a->polys, s) != hipSuccess)
      hr = from_status(kHipError);
    if (hr == HIPBFV_S_OK) hr = sync_stream(s);  // consts is a stack object: finish before it goes away
    g_buffers.put(dc, consts.size());
  }
  if (hr == HIPBFV_S_OK) hr = sync_stream(s);
  if (hr != HIPBFV_S_OK) {
    g_buffers.put(out, a->words);
    return hr;
  }
  g_buffers.put(a->dev, a->words);
  a->dev = out;
  a->rns = to_rns;
  return HIPBFV_S_OK;
}
long PolynomialArray_ToRns(void* h) HIPBFV_BEGIN
  PolyArrayObj* a = as<PolyArrayObj>(h, kMagicPolyArray);
  if (!a) return HIPBFV_E_POINTER;
  return polyarray_convert(a, true);
HIPBFV_END
long PolynomialArray_ToMultiprecision(void* h) HIPBFV_BEGIN
  PolyArrayObj* a = as<PolyArrayObj>(h, kMagicPolyArray);
  if (!a) return HIPBFV_E_POINTER;
  return polyarray_convert(a, false);
HIPBFV_END
// Drop the last prime: the new array keeps residues 0..kc-2 of every polynomial (RNS form)
long PolynomialArray_Drop(void* h, void** out) HIPBFV_BEGIN
  PolyArrayObj* a = as<PolyArrayObj>(h, kMagicPolyArray);
  if (!a || !out) return HIPBFV_E_POINTER;
  if (!a->reserved || a->kc < 2) return fail(HIPBFV_E_INVALIDARG, "no modulus to drop");
  const bool was_rns = a->rns;
  if (long hr = polyarray_convert(a, true)) return hr;
  std::unique_ptr<PolyArrayObj> b(new PolyArrayObj());
  long hr = polyarray_fill(b.get(), a->ctx, a->polys, a->kc - 1);
  if (hr == HIPBFV_S_OK && b->words) {
    hipStream_t s = thread_stream();
    const size_t row = (size_t)a->ctx->n() * sizeof(u64);
    if (hipMemcpy2DAsync(b->dev, b->kc * row, a->dev, a->kc * row, b->kc * row, a->polys, hipMemcpyDeviceToDevice, s) != hipSuccess)
      hr = from_status(kHipError);
    else
      hr = sync_stream(s);
  }
  if (!was_rns) polyarray_convert(a, false);
  if (hr != HIPBFV_S_OK) return hr;
  *out = b.release();
  return HIPBFV_S_OK;
HIPBFV_END

// r_i = floor(((q mod t) * m_i + (t + 1)/2) / t): what SEAL's multiply_add_plain_with_scaling_variant adds on top of
// floor(q/t) * m_i, so that c0 = floor(q/t)*m + r + ... exactly (logproof/src/bfv_statement.rs:159-160)
static void scaling_remainder(const Context& cx, const std::vector<u64>& m, PlainObj* r) {
  const u64 t = cx.t(), qt = cx.host().q_mod_t, half = (t + 1) >> 1;
  r->coeffs.assign(cx.n(), 0);
  for (size_t i = 0; i < m.size(); i++) r->coeffs[i] = (u64)(((unsigned __int128)qt * m[i] + half) / t);
}
// shared body: symmetric (sk) or public-key encryption of one plaintext, optionally exporting u, e, r
static long encrypt_one(EncryptorObj* e, PlainObj* p, CipherObj* c, bool symmetric, bool no_special, PolyArrayObj* u_dest, PolyArrayObj* e_dest,
                        PlainObj* r_dest, const RngSeed* fixed_seed) {
  const Context& cx = *e->ctx;
  const size_t n = cx.n(), words = cx.ct_words(2);
  if (symmetric ? !e->sk : !e->pk) return fail(HIPBFV_COR_E_INVALIDOPERATION, symmetric ? "secret key is not set" : "public key is not set");
  if (p->coeffs.size() > n) return fail(HIPBFV_E_INVALIDARG, "plain is not valid for encryption parameters");
  for (u64 v : p->coeffs)
    if (v >= cx.t()) return fail(HIPBFV_E_INVALIDARG, "plain is not valid for encryption parameters");
  u64 op = 0;
  RngSeed seed;
  if (fixed_seed) {
    seed = *fixed_seed;
  } else {
    std::lock_guard<std::mutex> g(e->mu);
    op = e->next_op++;
    seed = e->seed;
  }
  const u32 epolys = symmetric ? 1 : 2;
  if (u_dest)
    if (long hr = polyarray_fill(u_dest, e->ctx, 1, cx.K())) return hr;
  if (e_dest)
    if (long hr = polyarray_fill(e_dest, e->ctx, epolys, cx.K())) return hr;
  std::vector<u64> coeffs(n, 0);
  std::copy(p->coeffs.begin(), p->coeffs.end(), coeffs.begin());
  hipStream_t s = thread_stream();
  u64* pl = g_buffers.get(n);
  u64* out = g_buffers.get(words);
  if (!pl || !out) {
    g_buffers.put(pl, n);
    g_buffers.put(out, words);
    return from_status(kOutOfMemory);
  }
  long hr = HIPBFV_S_OK;
  int st = kOk;
  if (hipMemcpyAsync(pl, coeffs.data(), n * sizeof(u64), hipMemcpyHostToDevice, s) != hipSuccess)
    hr = from_status(kHipError);
  else if (symmetric)
    st = e->ev->encrypt_symmetric(pl, e->sk->dev, seed, op + 1, out, e_dest ? e_dest->dev : nullptr, s);
  else
    st = e->ev->encrypt_components(pl, e->pk->dev, seed, op, no_special, out, u_dest ? u_dest->dev : nullptr, e_dest ? e_dest->dev : nullptr, s);
  if (hr == HIPBFV_S_OK) hr = st ? from_status(st) : sync_stream(s);
  g_buffers.put(pl, n);
  if (hr != HIPBFV_S_OK) {
    g_buffers.put(out, words);
    return hr;
  }
  c->adopt(e->ctx, 2, out, words);
  if (r_dest) scaling_remainder(cx, p->coeffs, r_dest);
  return HIPBFV_S_OK;
}
long Encryptor_EncryptReturnComponents(void* h, void* plaintext, bool disable_special_modulus, void* destination, void* u_destination,
                                       void* e_destination, void* r_destination, void* pool) HIPBFV_BEGIN
  (void)pool;
  EncryptorObj* e = as<EncryptorObj>(h, kMagicEncryptor);
  PlainObj* p = as<PlainObj>(plaintext, kMagicPlain);
  CipherObj* c = as<CipherObj>(destination, kMagicCipher);
  PolyArrayObj* ud = as<PolyArrayObj>(u_destination, kMagicPolyArray);
  PolyArrayObj* ed = as<PolyArrayObj>(e_destination, kMagicPolyArray);
  PlainObj* rd = as<PlainObj>(r_destination, kMagicPlain);
  if (!e || !p || !c || !ud || !ed || !rd) return HIPBFV_E_POINTER;
  return encrypt_one(e, p, c, false, disable_special_modulus, ud, ed, rd, nullptr);
HIPBFV_END
long Encryptor_EncryptReturnComponentsSetSeed(void* h, void* plaintext, bool disable_special_modulus, void* destination, void* u_destination,
                                              void* e_destination, void* r_destination, void* seed, void* pool) HIPBFV_BEGIN
  (void)pool;
  EncryptorObj* e = as<EncryptorObj>(h, kMagicEncryptor);
  PlainObj* p = as<PlainObj>(plaintext, kMagicPlain);
  CipherObj* c = as<CipherObj>(destination, kMagicCipher);
  PolyArrayObj* ud = as<PolyArrayObj>(u_destination, kMagicPolyArray);
  PolyArrayObj* ed = as<PolyArrayObj>(e_destination, kMagicPolyArray);
  PlainObj* rd = as<PlainObj>(r_destination, kMagicPlain);
  if (!e || !p || !c || !ud || !ed || !rd || !seed) return HIPBFV_E_POINTER;
  const RngSeed sd = rng_seed_from_512(seed);  // the fork passes [u64; 8]: all 512 bits key the generator
  return encrypt_one(e, p, c, false, disable_special_modulus, ud, ed, rd, &sd);
HIPBFV_END
long Encryptor_EncryptSymmetric(void* h, void* plaintext, bool save_seed, void* destination, void* pool) HIPBFV_BEGIN
  (void)pool;
  (void)save_seed;  // seed-compressed ciphertexts are a serialisation option; the handle holds the expanded ciphertext
  EncryptorObj* e = as<EncryptorObj>(h, kMagicEncryptor);
  PlainObj* p = as<PlainObj>(plaintext, kMagicPlain);
  CipherObj* c = as<CipherObj>(destination, kMagicCipher);
  if (!e || !p || !c) return HIPBFV_E_POINTER;
  return encrypt_one(e, p, c, true, false, nullptr, nullptr, nullptr, nullptr);
HIPBFV_END
long Encryptor_EncryptSymmetricReturnComponents(void* h, void* plaintext, void* destination, void* e_destination, void* r_destination, void* pool) HIPBFV_BEGIN
  (void)pool;
  EncryptorObj* e = as<EncryptorObj>(h, kMagicEncryptor);
  PlainObj* p = as<PlainObj>(plaintext, kMagicPlain);
  CipherObj* c = as<CipherObj>(destination, kMagicCipher);
  PolyArrayObj* ed = as<PolyArrayObj>(e_destination, kMagicPolyArray);
  PlainObj* rd = as<PlainObj>(r_destination, kMagicPlain);
  if (!e || !p || !c || !ed || !rd) return HIPBFV_E_POINTER;
  return encrypt_one(e, p, c, true, false, nullptr, ed, rd, nullptr);
HIPBFV_END
long Encryptor_EncryptSymmetricReturnComponentsSetSeed(void* h, void* plaintext, void* destination, void* e_destination, void* r_destination,
                                                       void* seed, void* pool) HIPBFV_BEGIN
  (void)pool;
  EncryptorObj* e = as<EncryptorObj>(h, kMagicEncryptor);
  PlainObj* p = as<PlainObj>(plaintext, kMagicPlain);
  CipherObj* c = as<CipherObj>(destination, kMagicCipher);
  PolyArrayObj* ed = as<PolyArrayObj>(e_destination, kMagicPolyArray);
  PlainObj* rd = as<PlainObj>(r_destination, kMagicPlain);
  if (!e || !p || !c || !ed || !rd || !seed) return HIPBFV_E_POINTER;
  const RngSeed sd = rng_seed_from_512(seed);  // the fork passes [u64; 8]: all 512 bits key the generator
  return encrypt_one(e, p, c, true, false, nullptr, ed, rd, &sd);
HIPBFV_END

// ------------------------------------------------------------------ KeyGenerator (seal_fhe/src/key_generator.rs:20-200)
static std::shared_ptr<KeyBuffer> new_key_buffer(const std::shared_ptr<Context>& ctx, size_t words) {
  auto b = std::make_shared<KeyBuffer>();
  b->ctx = ctx;
  b->words = words;
  b->dev = g_buffers.get(words);
  return b->dev ? b : nullptr;
}
static long keygen_new(void* context, AsymKeyObj* existing, void** out, const RngSeed* fixed_seed = nullptr) {
  ContextObj* x = as<ContextObj>(context, kMagicContext);
  if (!x || !out) return HIPBFV_E_POINTER;
  if (existing && (!existing->key || existing->key->ctx.get() != x->ctx.get()))
    return fail(HIPBFV_E_INVALIDARG, "secret key is not valid for encryption parameters");
  std::unique_ptr<KeyGenObj> g(new KeyGenObj());
  g->ctx = x->ctx;
  g->ev.reset(new Evaluator(x->ctx.get()));
  if (fixed_seed)
    g->seed = *fixed_seed;
  else if (!os_seed(&g->seed))
    return fail(HIPBFV_E_UNEXPECTED, "no entropy available from the operating system (getrandom failed)");
  const size_t words = (size_t)x->ctx->KK() * x->ctx->n();
  g->sk_coeff = new_key_buffer(x->ctx, words);
  if (!g->sk_coeff) return from_status(kOutOfMemory);
  hipStream_t s = thread_stream();
  int st;
  if (existing) {  // recover the ternary polynomial: the inverse transform of every residue row
    g->sk = existing->key;
    st = hipMemcpyAsync(g->sk_coeff->dev, g->sk->dev, words * sizeof(u64), hipMemcpyDeviceToDevice, s) == hipSuccess ? kOk : kHipError;
    if (st == kOk) st = g->ev->ntt(g->sk_coeff->dev, x->ctx->KK(), x->ctx->KK(), true, s);
  } else {
    g->sk = new_key_buffer(x->ctx, words);
    if (!g->sk) return from_status(kOutOfMemory);
    st = g->ev->keygen_secret(g->seed, g->sk_coeff->dev, g->sk->dev, s);
  }
  if (st) return from_status(st);
  if (long hr = sync_stream(s)) return hr;
  *out = g.release();
  return HIPBFV_S_OK;
}
long KeyGenerator_Create1(void* context, void** out) HIPBFV_BEGIN return keygen_new(context, nullptr, out); HIPBFV_END
long hipbfv_KeyGenerator_CreateSeeded(void* context, uint64_t seed, void** out) HIPBFV_BEGIN
  const RngSeed sd = rng_seed_from_u64_for_tests(seed);  // TEST ONLY: 64 bits of entropy
  return keygen_new(context, nullptr, out, &sd);
HIPBFV_END
long KeyGenerator_Create2(void* context, void* secret_key, void** out) HIPBFV_BEGIN
  AsymKeyObj* k = as<AsymKeyObj>(secret_key, kMagicSecretKey);
  if (!k) return HIPBFV_E_POINTER;
  return keygen_new(context, k, out);
HIPBFV_END
long KeyGenerator_Destroy(void* h) HIPBFV_BEGIN
  KeyGenObj* g = as<KeyGenObj>(h, kMagicKeyGen);
  if (!g) return HIPBFV_E_POINTER;
  delete g;
  return HIPBFV_S_OK;
HIPBFV_END
long hipbfv_KeyGenerator_SetSeed(void* h, uint64_t seed) HIPBFV_BEGIN  // TEST ONLY (64 bits of entropy): reproducible keys; affects keys created afterwards
  KeyGenObj* g = as<KeyGenObj>(h, kMagicKeyGen);
  if (!g) return HIPBFV_E_POINTER;
  std::lock_guard<std::mutex> lk(g->mu);
  g->seed = rng_seed_from_u64_for_tests(seed);
  g->next_stream = 1;
  return HIPBFV_S_OK;
HIPBFV_END
long KeyGenerator_SecretKey(void* h, void** secret_key) HIPBFV_BEGIN
  KeyGenObj* g = as<KeyGenObj>(h, kMagicKeyGen);
  if (!g || !secret_key) return HIPBFV_E_POINTER;
  AsymKeyObj* k = new AsymKeyObj(kMagicSecretKey);
  k->key = g->sk;
  *secret_key = k;
  return HIPBFV_S_OK;
HIPBFV_END
static u64 take_stream(KeyGenObj* g, RngSeed* seed) {
  std::lock_guard<std::mutex> lk(g->mu);
  *seed = g->seed;
  return g->next_stream++;
}
long KeyGenerator_CreatePublicKey(void* h, bool save_seed, void** public_key) HIPBFV_BEGIN
  (void)save_seed;  // seed-compressed keys are a serialisation option; the handle always holds the expanded key
  KeyGenObj* g = as<KeyGenObj>(h, kMagicKeyGen);
  if (!g || !public_key) return HIPBFV_E_POINTER;
  auto buf = new_key_buffer(g->ctx, (size_t)2 * g->ctx->KK() * g->ctx->n());
  if (!buf) return from_status(kOutOfMemory);
  hipStream_t s = thread_stream();
  RngSeed seed;
  const u64 stream = take_stream(g, &seed);
  if (int st = g->ev->keygen_zero_encryptions(seed, stream, g->sk->dev, nullptr, buf->dev, 1, s)) return from_status(st);
  if (long hr = sync_stream(s)) return hr;
  AsymKeyObj* k = new AsymKeyObj(kMagicPublicKey);
  k->key = buf;
  *public_key = k;
  return HIPBFV_S_OK;
HIPBFV_END
static long keygen_kswitch_into(KeyGenObj* g, KeysObj* keys, u32 galois_elt) {
  const size_t words = g->ctx->key_words();
  u64* buf = g_buffers.get(words);
  if (!buf) return from_status(kOutOfMemory);
  hipStream_t s = thread_stream();
  RngSeed seed;
  const u64 stream = take_stream(g, &seed);
  int st = g->ev->keygen_kswitch(seed, stream, g->sk_coeff->dev, g->sk->dev, galois_elt, buf, s);
  long hr = st ? from_status(st) : sync_stream(s);
  if (hr != HIPBFV_S_OK) {
    g_buffers.put(buf, words);
    return hr;
  }
  const u32 index = galois_elt ? (galois_elt - 1) >> 1 : 0;
  auto it = keys->keys.find(index);
  if (it != keys->keys.end()) g_buffers.put(it->second, words);
  keys->keys[index] = buf;
  return HIPBFV_S_OK;
}
long KeyGenerator_CreateRelinKeys(void* h, bool save_seed, void** relin_keys) HIPBFV_BEGIN
  (void)save_seed;
  KeyGenObj* g = as<KeyGenObj>(h, kMagicKeyGen);
  if (!g || !relin_keys) return HIPBFV_E_POINTER;
  if (g->ctx->KK() < 2) return fail(HIPBFV_COR_E_INVALIDOPERATION, "keyswitching is not supported by the context");
  std::unique_ptr<KeysObj> k(new KeysObj());
  k->ctx = g->ctx;
  if (long hr = keygen_kswitch_into(g, k.get(), 0)) return hr;
  *relin_keys = k.release();
  return HIPBFV_S_OK;
HIPBFV_END
long KeyGenerator_CreateGaloisKeysFromElts(void* h, uint64_t count, uint32_t* galois_elts, bool save_seed, void** galois_keys) HIPBFV_BEGIN
  (void)save_seed;
  KeyGenObj* g = as<KeyGenObj>(h, kMagicKeyGen);
  if (!g || !galois_keys || (count && !galois_elts)) return HIPBFV_E_POINTER;
  if (g->ctx->KK() < 2) return fail(HIPBFV_COR_E_INVALIDOPERATION, "keyswitching is not supported by the context");
  if (!g->ctx->batching()) return fail(HIPBFV_COR_E_INVALIDOPERATION, "encryption parameters do not support batching");
  std::unique_ptr<KeysObj> k(new KeysObj());
  k->ctx = g->ctx;
  for (uint64_t i = 0; i < count; i++) {
    const u32 elt = galois_elts[i];
    if (!(elt & 1) || elt >= 2 * g->ctx->n()) return fail(HIPBFV_E_INVALIDARG, "Galois element is not valid");
    if (long hr = keygen_kswitch_into(g, k.get(), elt)) return hr;
  }
  *galois_keys = k.release();
  return HIPBFV_S_OK;
HIPBFV_END
// SEAL KeyGenerator::create_galois_keys(steps): one key per rotation step, GaloisTool::get_elts_from_steps (step 0 = the
// column rotation).  The seal_fhe crate does not call it; exported so that the whole KeyGenerator_* family links.
long KeyGenerator_CreateGaloisKeysFromSteps(void* h, uint64_t count, int* steps, bool save_seed, void** galois_keys) HIPBFV_BEGIN
  KeyGenObj* g = as<KeyGenObj>(h, kMagicKeyGen);
  if (!g || !galois_keys || (count && !steps)) return HIPBFV_E_POINTER;
  std::vector<uint32_t> elts;
  for (uint64_t i = 0; i < count; i++) {
    const u32 elt = g->ev->galois_elt_from_step(steps[i]);
    if (!elt) return fail(HIPBFV_E_INVALIDARG, "step count too large");
    elts.push_back(elt);
  }
  return KeyGenerator_CreateGaloisKeysFromElts(h, elts.size(), elts.data(), save_seed, galois_keys);
HIPBFV_END
// SEAL GaloisTool::get_elts_all: the column rotation 2N-1 and 3^(+-2^i) for every power-of-two row rotation
long KeyGenerator_CreateGaloisKeysAll(void* h, bool save_seed, void** galois_keys) HIPBFV_BEGIN
  KeyGenObj* g = as<KeyGenObj>(h, kMagicKeyGen);
  if (!g || !galois_keys) return HIPBFV_E_POINTER;
  const u64 m = 2 * (u64)g->ctx->n();
  int logn = 0;
  while ((1u << logn) < g->ctx->n()) logn++;
  std::vector<uint32_t> elts;
  elts.push_back((uint32_t)(m - 1));
  u64 pos = 3, neg = 1;
  for (int i = 0; i < 6; i++) neg = neg * (2 - 3 * neg);
  neg &= m - 1;
  for (int i = 0; i < logn - 1; i++) {
    elts.push_back((uint32_t)pos);
    pos = (pos * pos) & (m - 1);
    elts.push_back((uint32_t)neg);
    neg = (neg * neg) & (m - 1);
  }
  return KeyGenerator_CreateGaloisKeysFromElts(h, elts.size(), elts.data(), save_seed, galois_keys);
HIPBFV_END

}  // extern "C"
#pragma GCC visibility pop
