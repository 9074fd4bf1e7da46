// sunscreen_amd/csrc/evaluator.cpp -- see evaluator.hpp.
#include "evaluator.hpp"
#include "nttshape.hpp"

#include <algorithm>
#include <cstdlib>

namespace hipbfv {

#define HB_CHECK(expr)                         \
  do {                                         \
    hipError_t e__ = (expr);                   \
    if (e__ != hipSuccess) return kHipError;   \
  } while (0)

// ------------------------------------------------------------------ ScratchPool

ScratchPool::~ScratchPool() { trim(); }

void ScratchPool::trim() {
  std::lock_guard<std::mutex> g(mu_);
  for (auto it = blocks_.begin(); it != blocks_.end();) {
    if (!it->busy) {
      (void)hipEventSynchronize(it->ev);
      (void)hipEventDestroy(it->ev);
      (void)hipFree(it->ptr);
      it = blocks_.erase(it);
    } else {
      ++it;
    }
  }
}

void* ScratchPool::acquire(size_t bytes, hipStream_t s) {
  if (bytes == 0) bytes = 256;
  std::lock_guard<std::mutex> g(mu_);
  Block* best = nullptr;
  for (auto& b : blocks_)
    if (!b.busy && b.bytes >= bytes && (!best || b.bytes < best->bytes)) best = &b;
  if (best) {
    best->busy = true;
    // a block last used on another stream may still be read by kernels queued there
    if (best->last != s) (void)hipStreamWaitEvent(s, best->ev, 0);
    return best->ptr;
  }
  Block nb{};
  if (hipMalloc(&nb.ptr, bytes) != hipSuccess) {
    // free cached blocks and retry once
    for (auto it = blocks_.begin(); it != blocks_.end();) {
      if (!it->busy) {
        (void)hipEventSynchronize(it->ev);
        (void)hipEventDestroy(it->ev);
        (void)hipFree(it->ptr);
        it = blocks_.erase(it);
      } else {
        ++it;
      }
    }
    if (hipMalloc(&nb.ptr, bytes) != hipSuccess) return nullptr;
  }
  nb.bytes = bytes;
  nb.last = s;
  nb.busy = true;
  if (hipEventCreateWithFlags(&nb.ev, hipEventDisableTiming) != hipSuccess) {
    (void)hipFree(nb.ptr);
    return nullptr;
  }
  blocks_.push_back(nb);
  return nb.ptr;
}

void ScratchPool::release(void* p, hipStream_t s) {
  std::lock_guard<std::mutex> g(mu_);
  for (auto& b : blocks_) {
    if (b.ptr == p) {
      b.last = s;
      (void)hipEventRecord(b.ev, s);
      b.busy = false;
      return;
    }
  }
}

// ------------------------------------------------------------------ PinnedPool

PinnedPool::~PinnedPool() {
  for (auto& b : blocks_) {
    if (b.pending) (void)hipEventSynchronize(b.ev);
    (void)hipEventDestroy(b.ev);
    (void)hipHostFree(b.ptr);
  }
}

void* PinnedPool::acquire(size_t bytes) {
  if (bytes == 0) bytes = 256;
  std::unique_lock<std::mutex> g(mu_);
  Block* best = nullptr;
  for (auto& b : blocks_)
    if (!b.busy && b.bytes >= bytes && (!best || b.bytes < best->bytes)) best = &b;
  if (best) {
    best->busy = true;
    const bool wait = best->pending;
    hipEvent_t ev = best->ev;
    void* p = best->ptr;
    best->pending = false;
    g.unlock();
    // the previous copy out of the block (usually long finished) must be over before the host writes it again
    if (wait && hipEventSynchronize(ev) != hipSuccess) return nullptr;
    return p;
  }
  Block nb{};
  const size_t want = std::max<size_t>(bytes, (size_t)64 << 10);
  if (hipHostMalloc(&nb.ptr, want, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  if (hipEventCreateWithFlags(&nb.ev, hipEventDisableTiming) != hipSuccess) {
    (void)hipHostFree(nb.ptr);
    return nullptr;
  }
  nb.bytes = want;
  nb.busy = true;
  nb.pending = false;
  blocks_.push_back(nb);
  return nb.ptr;
}

void PinnedPool::release(void* p, hipStream_t s) {
  std::lock_guard<std::mutex> g(mu_);
  for (auto& b : blocks_) {
    if (b.ptr == p) {
      b.pending = hipEventRecord(b.ev, s) == hipSuccess;
      if (!b.pending) (void)hipStreamSynchronize(s);  // no event: drain the stream instead (the copy must not outlive the block)
      b.busy = false;
      return;
    }
  }
}

namespace {
NttPlan make_plan(u32 div, const std::vector<u32>& mods) {
  NttPlan pl{};
  pl.div = div;
  pl.period = (u32)mods.size();
  for (size_t i = 0; i < mods.size(); i++) pl.mod[i] = (unsigned char)mods[i];
  return pl;
}
}  // namespace

// ------------------------------------------------------------------ Profiler

const char* kernel_name(int id) {
  static const char* names[kKernCount] = {"ntt_fwd",   "ntt_inv",    "behz_extend", "tensor",  "behz_floor_sk", "ks_decompose",
                                          "ks_mac",    "ks_moddown", "galois",      "eltwise", "plain",
                                          "ks_head",    "ks_mid",      "ks_tail", "mul_head",      "mul_mid",
                                          "mul_tail"};
  return id >= 0 && id < kKernCount ? names[id] : "?";
}

Profiler::~Profiler() {
  for (auto& r : recs_) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  for (auto e : free_) (void)hipEventDestroy(e);
}

hipEvent_t Profiler::get_event() {
  if (!free_.empty()) {
    hipEvent_t e = free_.back();
    free_.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

void Profiler::begin(int id, size_t units_, hipStream_t s) {
  if (!enabled) return;
  std::lock_guard<std::mutex> g(mu_);
  Rec r{id, units_, get_event(), get_event()};
  (void)hipEventRecord(r.a, s);
  recs_.push_back(r);
}

void Profiler::end(hipStream_t s) {
  if (!enabled) return;
  std::lock_guard<std::mutex> g(mu_);
  if (!recs_.empty()) (void)hipEventRecord(recs_.back().b, s);
}

void Profiler::collect() {
  std::lock_guard<std::mutex> g(mu_);
  for (auto& r : recs_) {
    (void)hipEventSynchronize(r.b);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      total_ms[r.id] += ms;
      launches[r.id] += 1;
      units[r.id] += r.units;
    }
    free_.push_back(r.a);
    free_.push_back(r.b);
  }
  recs_.clear();
}

void Profiler::reset() {
  collect();
  for (int i = 0; i < kKernCount; i++) total_ms[i] = 0, launches[i] = 0, units[i] = 0;
}

// launch `expr` bracketed by profiler events
#define HB_LAUNCH(id, units, expr)  \
  do {                              \
    prof_.begin(id, units, s);      \
    hipError_t e__ = (expr);        \
    prof_.end(s);                   \
    if (e__ != hipSuccess) return kHipError; \
  } while (0)

// ------------------------------------------------------------------ Evaluator

Evaluator::Evaluator(Context* ctx) : ctx_(ctx) {
  const DevCtx& h = ctx_->host();
  const size_t R = h.K + h.S;
  const size_t per_op = (size_t)(4 * R + 3 * R + 3 * h.K + (size_t)h.KK * h.K + 2 * h.KK) * h.n * sizeof(u64);
  // up to 32 GiB of scratch per chunk (of 288 GB): launches of 1024 ops at every degree up to N = 16384.  Measured at N = 16384
  // (r03, one box): chunks of 256 / 512 / 1024 ops give 49.6 / 50.2 / 51.3 K mul+relin/s -- a longer walk over one key slice
  // per XCD in ks_mid (5.86 -> 5.48 ms) and fewer launch tails; the 8 GiB cap of rounds 1-2 meant 292 ops there.
  // r05: the cap on ops per chunk is 4096 (was 1024): at N <= 8192 a chunk is then limited by the cap, at N = 16384 by the 32 GiB of
  // scratch (~1100 ops).  Interleaved on one box (profiles/r05_chunk_ab.txt): n = 8192, batch 4096: 330.0 K -> 331.8 K (2048 per chunk)
  // -> 332.2 K (4096); n = 16384, batch 2048, 1024 -> 2048 per chunk: 58.1 K -> 59.2 K -- fewer launch tails, as r03 found below 1024.
  size_t c = ((size_t)32 << 30) / per_op;
  if (const char* env = std::getenv("HIPBFV_CHUNK_OPS")) c = (size_t)std::strtoull(env, nullptr, 10);
  chunk_ops_ = std::max<size_t>(1, std::min<size_t>(c, 4096));  // <= 32 GiB of scratch AND <= 4096 ops (28 GiB at N = 8192, K = 4)
  if (const char* env = std::getenv("HIPBFV_NO_SPLIT_KS")) split_ks_ = env[0] != '1';
  if (const char* env = std::getenv("HIPBFV_NO_SPLIT_MUL")) split_mul_ = env[0] != '1';
  if (const char* env = std::getenv("HIPBFV_NO_FUSED_TAIL")) fuse_mulrelin_ = env[0] != '1';
  if (const char* env = std::getenv("HIPBFV_NO_FUSED_HEAD")) fuse_head_ = env[0] != '1';
  if (const char* env = std::getenv("HIPBFV_NO_SMALL_BATCH")) small_batch_ = env[0] != '1';
  if (const char* env = std::getenv("HIPBFV_NO_FUSED_GALOIS")) fuse_galois_ = env[0] != '1';
  {
    // The coefficient-parallel split kernels address one batch item's rows through a buffer descriptor with a 32-bit offset
    // (kernels_split.hip buf_row): every span they put under one descriptor -- T (KK * K rows), ext (4 R), D (3 R), ACC (2 KK) of ONE
    // item -- must stay below 4 GiB.  63 MB at most today (N = 32768, K = 15); a context that ever exceeded it takes the
    // whole-polynomial kernels instead of wrapping silently.
    const size_t rows = std::max<size_t>(std::max<size_t>((size_t)h.KK * h.K, 4 * R), 2 * (size_t)h.KK);
    if (rows * h.n * sizeof(u64) >= ((size_t)1 << 32)) split_ks_ = split_mul_ = false;
  }
  if (hipMalloc((void**)&status_dev_, 256) == hipSuccess)
    (void)hipMemset(status_dev_, 0xFF, 256);
  else {
    status_dev_ = nullptr;
    (void)hipGetLastError();
  }
}

namespace {
thread_local u32* tls_watch = nullptr;
}
WatchScope::WatchScope(u32* status_dev) : prev(tls_watch) { tls_watch = status_dev; }
WatchScope::~WatchScope() { tls_watch = prev; }
u32* Evaluator::watch_status() { return tls_watch; }

Evaluator::~Evaluator() {
  if (status_dev_) (void)hipFree(status_dev_);
}

int Evaluator::note_result(const u64* ct, u32 size, u32 residues, size_t count, hipStream_t s) {
  u32* status = tls_watch;
  if (!status || size < 2 || !count) return kOk;
  const size_t poly = (size_t)residues * ctx_->n();
  for (size_t off = 0; off < count; off += (size_t)1 << 30) {
    const size_t c = std::min<size_t>((size_t)1 << 30, count - off);
    HB_CHECK(launch_transparent_watch(ct + off * size * poly, size * poly, poly, (u32)off, status, c, s));
  }
  return kOk;
}

int Evaluator::take_status(u32* status_dev, u32* first_bad, hipStream_t s) {
  if (!status_dev) return kOutOfMemory;
  *first_bad = 0xFFFFFFFFu;
  HB_CHECK(hipMemcpyAsync(first_bad, status_dev, sizeof(u32), hipMemcpyDeviceToHost, s));
  HB_CHECK(hipMemsetAsync(status_dev, 0xFF, sizeof(u32), s));
  HB_CHECK(hipStreamSynchronize(s));
  return kOk;
}

u32 Evaluator::galois_elt_from_step(int step) const {
  const u32 n = ctx_->n(), m = 2 * n;
  if (step == 0) return m - 1;
  const u32 pos = (u32)(step < 0 ? -(long)step : (long)step);
  if (pos >= (n >> 1)) return 0;
  const u32 e = step < 0 ? (n >> 1) - pos : pos;
  u64 g = 1;
  for (u32 i = 0; i < e; i++) g = (g * 3) & (m - 1);
  return (u32)g;
}

// A FEW ciphertexts (a handle-level call, a small combined batch) are latency-bound: the head / tail kernels of the split
// pipelines give each thread eight (four) coefficients and a few workgroups per ciphertext, the whole-polynomial pipelines
// spread the same work over many more, shorter workgroups.  tools/small_batch_sweep.py (time per call, split / whole, us):
//   N = 8192:  multiply 1: 70 / 59, 16: 113 / 107, 32: 120 / 166;  multiply+relinearize fused 1: 147 (unfused 110), 16: 173 / 174
//   N = 16384: multiply 1: 118 / 110, 8: 169 / 203;  relinearize 1: 117 / 74, 8: 138 / 141;  rotation 1: 123 / 80, 8: 147 / 151
//   N = 4096:  the split pipelines win at every count.
// The results are the same bits either way (test_split_and_whole_polynomial_paths_agree).
bool Evaluator::few_for_split_mul(size_t count) const {
  const u32 logn = ctx_->host().logn;
  return small_batch_ && (logn == 13 ? count <= 16 : logn == 14 ? count <= 4 : false);
}
bool Evaluator::few_for_split_ks(size_t count) const { return small_batch_ && ctx_->host().logn == 14 && count <= 4; }
bool Evaluator::few_for_fused(size_t count) const {
  const u32 logn = ctx_->host().logn;
  return small_batch_ && (logn == 13 ? count <= 8 : logn == 14 ? count <= 4 : false);
}

int Evaluator::ntt(u64* data, size_t polys, u32 nprimes, bool inverse, hipStream_t s) {
  const DevCtx& h = ctx_->host();
  if (nprimes == 0 || nprimes > h.KK) return kInvalidArg;
  if (h.logn > 15) return kUnsupported;
  std::vector<u32> mods(nprimes);
  for (u32 i = 0; i < nprimes; i++) mods[i] = i;
  const NttPlan plan = make_plan(1, mods);
  const size_t step = (65535 / nprimes) * nprimes;
  for (size_t off = 0; off < polys; off += step) {
    const size_t cnt = std::min(step, polys - off);
    HB_LAUNCH((inverse ? kKernNttInv : kKernNttFwd), cnt, launch_ntt(ctx_->dev(), (inverse ? h.tw_inv : h.tw_fwd), h.logn, data + off * h.n, cnt, plan, inverse, 0, s));
  }
  return kOk;
}

int Evaluator::multiply(const u64* a, u32 sa, const u64* b, u32 sb, u64* out, size_t count, hipStream_t s, bool watch) {
  const DevCtx& h = ctx_->host();
  if (sa < 2 || sb < 2 || sa + sb > 16) return kInvalidArg;
  if (h.logn > 15) return kUnsupported;
  const u32 n = h.n, K = h.K, S = h.S, R = K + S, sd = sa + sb - 1;
  // the per-coefficient kernels are instantiated for KMAX data primes and KMAX + 2 auxiliary primes
  const u32 kneed = std::max(K, S > 2 ? S - 2 : 0u);
  const bool split = split_mul_ && sa == 2 && sb == 2 && (kneed <= 4 || (kneed <= 8 && h.aux_f64)) && h.logn >= 12 && h.logn <= 14 &&
                     !few_for_split_mul(count);
  // (the split kernels put the items on grid z; only the whole-polynomial launches count residue polynomials on one grid axis.
  // r06: the 3 x 54-bit multiply of 4096 items ran as 1820 + 1820 + 456 under the shared limit)
  const size_t chunk = std::max<size_t>(1, std::min<size_t>(chunk_ops_, split ? 65535 : 65535 / (R * (sa + sb))));
  const size_t ext_words = (size_t)(sa + sb) * R * n, d_words = (size_t)sd * R * n;
  const size_t cc = std::min(chunk, count);  // a handle-level call (count = 1) reserves one op's scratch, not a chunk's
  ScratchGuard sg(pool_, cc * (ext_words + d_words) * sizeof(u64), s);
  if (!sg.p) return kOutOfMemory;
  u64* ext = (u64*)sg.p;
  u64* D = ext + cc * ext_words;
  std::vector<u32> mods;
  for (u32 i = 0; i < K; i++) mods.push_back(i);
  for (u32 j = 0; j < S; j++) mods.push_back(h.KK + j);
  const NttPlan plan = make_plan(1, mods);
  // x * x (Evaluator_Square, a program node with one operand twice): the split kernels extend and transform x once
  const bool square = split && a == b;
  for (size_t off = 0; off < count; off += chunk) {
    const size_t c = std::min(chunk, count - off);
    if (split) {
      // head / middle / tail split transforms (kernels_split.hip): 3 launches instead of 5, 40 % less HBM traffic
      HB_LAUNCH(kKernMulHead, c * (square ? 2 : 4), launch_mul_head(ctx_->dev(), h.tw_fwd, h.logn, h.aux_f64 != 0, h.aux_f64 ? (int)h.pack_mul | (h.conv_grid == 1 ? 4 : 0) : (h.aux_mixed ? 1 : 0), kneed, a + off * 2 * K * n, b + off * 2 * K * n, ext, c, s, square ? 2u : 4u));
      HB_LAUNCH(kKernMulMid, c, launch_mul_mid(ctx_->dev(), h.tw_fwd, h.tw_inv, h.logn, ctx_->dev()->mid_res_dp, h.mid_ndp, ctx_->dev()->mid_res_d, h.mid_nd, ctx_->dev()->mid_res_i, h.mid_ni, ext, D, c, s, square));
      HB_LAUNCH(kKernMulTail, c * 3, launch_mul_tail(ctx_->dev(), h.tw_inv, h.logn, h.aux_f64 != 0, h.aux_f64 ? (int)h.pack_mul : (h.aux_mixed ? 1 : 0), h.conv_grid != 0, kneed, D, out + off * 3 * K * n, c, s));
      continue;
    }
    HB_LAUNCH(kKernBehzExtend, c * (sa + sb), launch_behz_extend(ctx_->dev(), n, kneed, a + off * sa * K * n, sa, b + off * sb * K * n, sb, c, ext, s));
    HB_LAUNCH(kKernNttFwd, c * (sa + sb) * R, launch_ntt(ctx_->dev(), h.tw_fwd, h.logn, ext, c * (sa + sb) * R, plan, false, 0, s));
    HB_LAUNCH(kKernTensor, c, launch_tensor(ctx_->dev(), n, R, ext, sa, sb, D, c, s));
    HB_LAUNCH(kKernNttInv, c * sd * R, launch_ntt(ctx_->dev(), h.tw_inv, h.logn, D, c * sd * R, plan, true, 1, s));
    HB_LAUNCH(kKernBehzFloorSk, c * sd, launch_behz_floor_sk(ctx_->dev(), n, kneed, D, out + off * sd * K * n, c * sd, s));
  }
  return watch ? note_result(out, sd, K, count, s) : (int)kOk;
}

size_t Evaluator::ks_scratch_words() const {
  const DevCtx& h = ctx_->host();
  return ((size_t)h.KK * h.K + 2 * (size_t)h.KK) * h.n;
}

Evaluator::KeyMapLease::~KeyMapLease() {
  if (!ev) return;
  if (host) ev->pinned_.release(host, s);
  if (dev) ev->pool_.release(dev, s);
}

// Device tables of a per-item key selection (KeySel): the key pointer table and, for every chunk of the call, the chunk's items
// sorted by key index (stable: items of one key keep their order).  One pinned staging block and one H2D copy per call; both
// buffers go back to their pools behind the call's last launch.
int Evaluator::stage_keymap(const KeySel& sel, size_t count, size_t chunk, hipStream_t s, KeyMapLease& lease) {
  if (!sel.per_item()) return kOk;
  if (!sel.nkeys || !sel.index || !sel.period || !count || !chunk) return kInvalidArg;
  for (u32 k = 0; k < sel.nkeys; k++)
    if (!sel.keys[k]) return kNoKey;
  for (size_t i = 0; i < sel.period; i++)
    if (sel.index[i] >= sel.nkeys) return kInvalidArg;
  const size_t tab_bytes = ((size_t)sel.nkeys * sizeof(u64*) + 15) & ~(size_t)15;
  const size_t bytes = tab_bytes + count * sizeof(uint2);
  lease.ev = this;
  lease.s = s;
  lease.host = pinned_.acquire(bytes);
  // (one size class up to 128 K items: the small block is found again by every later call instead of a chunk-sized one)
  lease.dev = pool_.acquire(std::max<size_t>(bytes, (size_t)1 << 20), s);
  if (!lease.host || !lease.dev) return kOutOfMemory;
  const u64** tab = reinterpret_cast<const u64**>(lease.host);
  for (u32 k = 0; k < sel.nkeys; k++) tab[k] = sel.keys[k];
  uint2* order = reinterpret_cast<uint2*>(static_cast<unsigned char*>(lease.host) + tab_bytes);
  std::vector<u32> start(sel.nkeys + 1);
  for (size_t off = 0; off < count; off += chunk) {
    const size_t c = std::min(chunk, count - off);
    // counting sort of the chunk's items by key index
    std::fill(start.begin(), start.end(), 0u);
    for (size_t i = 0; i < c; i++) start[sel.index[(sel.first + off + i) % sel.period] + 1]++;
    for (u32 k = 0; k < sel.nkeys; k++) start[k + 1] += start[k];
    for (size_t i = 0; i < c; i++) {
      const u32 k = sel.index[(sel.first + off + i) % sel.period];
      order[off + start[k]++] = make_uint2((u32)i, k);
    }
  }
  HB_CHECK(hipMemcpyAsync(lease.dev, lease.host, bytes, hipMemcpyHostToDevice, s));
  lease.km.keys = reinterpret_cast<const u64* const*>(lease.dev);
  lease.km.order = reinterpret_cast<const uint2*>(static_cast<unsigned char*>(lease.dev) + tab_bytes);
  return kOk;
}

bool Evaluator::ks_split_for(size_t count) const {
  const DevCtx& h = ctx_->host();
  // (N = 32768 [r06]: integer-policy key primes only -- kernels_split.hip KS_DISPATCH; the multiply keeps the whole-polynomial kernels there)
  return split_ks_ && h.logn >= 12 && (h.logn <= 14 || (h.logn == 15 && h.ks_nd + h.ks_ndp == 0)) && h.ks_split_ok && !few_for_split_ks(count);
}

// out2[op] = base[op] (masked) + modDown( sum_J NTT(target_J) (.) key[J] ); scratch >= count * ks_scratch_words()
int Evaluator::key_switch(const u64* target, size_t tstride, const u64* key, const u64* base, size_t bstride, u32 base_mask,
                          u64* out2, size_t count, u64* scratch, hipStream_t s, const u64* extra, KeyMap km, u32 ginv) {
  const DevCtx& h = ctx_->host();
  const u32 n = h.n, K = h.K, KK = h.KK;
  u64* T = scratch;
  u64* ACC = scratch + count * (size_t)KK * K * n;
  std::vector<u32> mods;
  for (u32 i = 0; i < KK; i++) mods.push_back(i);
  const bool split_ok = ks_split_for(count);
  if (ginv && !split_ok) return kInvalidArg;  // (apply_galois rotates into a copy for the whole-polynomial kernels)
  if (split_ok) {
    // head / middle / tail split transforms (kernels_split.hip): 3 launches (4 when FP64- and integer-policy key primes are
    // mixed: one middle kernel per policy), no whole-polynomial NTT round trips
    const bool mixed = h.ks_ni != 0;
    HB_LAUNCH(kKernKsHead, count, launch_ks_head(ctx_->dev(), h.tw_fwd, h.logn, (int)h.pack_ks, mixed, K, target, tstride, T, count, s, ginv));
    HB_LAUNCH(kKernKsMid, count, launch_ks_mid(ctx_->dev(), h.tw_fwd, h.tw_inv, h.logn, h, T, key, ACC, count, s, km));
    HB_LAUNCH(kKernKsTail, count, launch_ks_tail(ctx_->dev(), h.tw_inv, h.logn, (int)h.pack_ks, mixed, ACC, base, bstride, base_mask, extra, out2, count, s, ginv));
    return kOk;
  }
  HB_LAUNCH(kKernKsDecompose, count, launch_ks_decompose(ctx_->dev(), n, K, target, tstride, T, count, s));
  HB_LAUNCH(kKernNttFwd, count * KK * K, launch_ntt(ctx_->dev(), h.tw_fwd, h.logn, T, count * KK * K, make_plan(K, mods), false, 0, s));
  HB_LAUNCH(kKernKsMac, count, launch_ks_mac(ctx_->dev(), n, KK, T, key, ACC, count, s, km));
  HB_LAUNCH(kKernNttInv, count * 2 * KK, launch_ntt(ctx_->dev(), h.tw_inv, h.logn, ACC, count * 2 * KK, make_plan(1, mods), true, 0, s));
  HB_LAUNCH(kKernKsModdown, count, launch_ks_moddown(ctx_->dev(), n, ACC, base, bstride, base_mask, extra, out2, count, s));
  return kOk;
}

// addend (optional): ciphertexts u64[count][2][K][N] added to the results inside the last kernel (a fused Add node)
int Evaluator::relinearize(const u64* ct3, const KeySel& rk, u64* out2, size_t count, hipStream_t s, const u64* addend, bool watch) {
  const DevCtx& h = ctx_->host();
  if (h.KK < 2 || !rk.present()) return kNoKey;
  if (h.logn > 15) return kUnsupported;
  const u32 n = h.n, K = h.K;
  const size_t chunk = std::max<size_t>(1, std::min<size_t>(chunk_ops_, 65535 / ((size_t)h.KK * K)));
  KeyMapLease kl;
  if (int rc = stage_keymap(rk, count, chunk, s, kl)) return rc;
  ScratchGuard sg(pool_, std::min(chunk, count) * ks_scratch_words() * sizeof(u64), s);
  if (!sg.p) return kOutOfMemory;
  const size_t cs = (size_t)3 * K * n;
  for (size_t off = 0; off < count; off += chunk) {
    const size_t c = std::min(chunk, count - off);
    const u64* ct = ct3 + off * cs;
    int rc = key_switch(ct + (size_t)2 * K * n, cs, rk.key, ct, cs, 3u, out2 + off * 2 * K * n, c, (u64*)sg.p, s, addend ? addend + off * 2 * K * n : nullptr,
                        kl.at(off));
    if (rc) return rc;
  }
  return watch ? note_result(out2, 2, K, count, s) : (int)kOk;
}

// the all-FP64 fused multiply + relinearize (every SEAL default set up to N = 16384) would run for a batch of `count`
bool Evaluator::member_tail_ok(size_t count) const {
  const DevCtx& h = ctx_->host();
  const u32 kneed = std::max(h.K, h.S > 2 ? h.S - 2 : 0u);
  return fuse_mulrelin_ && split_mul_ && split_ks_ && h.aux_f64 && h.ks_split_ok && h.ks_ni == 0 && kneed <= 8 && h.logn >= 12 && h.logn <= 14 && !few_for_fused(count);
}

int Evaluator::multiply_relin(const u64* a, const u64* b, const KeySel& rk, u64* out2, size_t count, hipStream_t s, const u64* addend, const MemberTail* members,
                              u32 per, const MemberHead* heads, bool heads_square) {
  const DevCtx& h = ctx_->host();
  if (h.KK < 2 || !rk.present()) return kNoKey;
  if (members && (!member_tail_ok(count) || !per || addend)) return kInvalidArg;
  if (heads && !members) return kInvalidArg;
  const u32 n = h.n, K = h.K, KK = h.KK, S = h.S, R = K + S;
  const size_t cs = (size_t)3 * K * n, c2 = (size_t)2 * K * n;
  const u32 kneed = std::max(K, S > 2 ? S - 2 : 0u);
  // Fused pipeline (all-FP64 contexts: every SEAL default set up to N = 16384): six launches; the product's c0 and c1 are
  // formed inside the last one (mulrelin_tail_kernel) and only c2 -- the key-switch target -- is written by mul_tail.
  const bool fused_d = member_tail_ok(count);
  // r06: the same five launches for MIXED contexts (integer-policy data / key primes beside the library's FP64 auxiliary base: the
  // 3 x 54-bit set): mulrelin_head_mixed / mulrelin_tail_mixed, 8-byte rows, one or two middle launches per policy
  const bool fused_m = !fused_d && fuse_mulrelin_ && fuse_head_ && split_mul_ && split_ks_ && !h.aux_f64 && h.aux_mixed && h.ks_split_ok && kneed <= 4 &&
                       h.logn >= 12 && h.logn <= 14 && !lane_split((int)h.logn) && !few_for_fused(count);
  const bool fused = fused_d || fused_m;
  const bool square = fused && (heads ? heads_square : a == b);  // x * x: the head extends and the middle kernel transforms x once
  if (fused) {
    const size_t ext_words = (size_t)4 * R * n, d_words = (size_t)3 * R * n, t_words = (size_t)KK * K * n, acc_words = (size_t)2 * KK * n, c2_words = (size_t)K * n;
    const size_t per_op = ext_words + d_words + t_words + acc_words + c2_words;
    // every kernel of this path puts the ops on grid z and its residue / block count on grid x: no 65535 limit but z's, which
    // the 4096-op cap of chunk_ops_ is far below (the limits of the whole-polynomial launches made N = 16384 run 910 + 114)
    const size_t chunk = std::max<size_t>(1, std::min<size_t>(chunk_ops_, 65535));
    KeyMapLease kl;
    if (int rc = stage_keymap(rk, count, chunk, s, kl)) return rc;
    ScratchGuard sg(pool_, std::min(chunk, count) * per_op * sizeof(u64), s);
    if (!sg.p) return kOutOfMemory;
    const size_t cc = std::min(chunk, count);
    u64* ext = (u64*)sg.p;
    u64* D = ext + cc * ext_words;
    u64* T = D + cc * d_words;
    u64* ACC = T + cc * t_words;
    u64* C2 = ACC + cc * acc_words;
    for (size_t off = 0; off < count; off += chunk) {
      const size_t c = std::min(chunk, count - off);
      if (fused_m) {
        HB_LAUNCH(kKernMulHead, c * (square ? 2 : 4), launch_mul_head(ctx_->dev(), h.tw_fwd, h.logn, false, 1, kneed, a + off * c2, b + off * c2, ext, c, s, square ? 2u : 4u));
        HB_LAUNCH(kKernMulMid, c, launch_mul_mid(ctx_->dev(), h.tw_fwd, h.tw_inv, h.logn, ctx_->dev()->mid_res_dp, h.mid_ndp, ctx_->dev()->mid_res_d, h.mid_nd, ctx_->dev()->mid_res_i, h.mid_ni, ext, D, c, s, square));
        HB_LAUNCH(kKernKsHead, c, launch_mulrelin_head_mixed(ctx_->dev(), h.tw_inv, h.tw_fwd, h.logn, D, T, c, s));
        HB_LAUNCH(kKernKsMid, c, launch_ks_mid(ctx_->dev(), h.tw_fwd, h.tw_inv, h.logn, h, T, rk.key, ACC, c, s, kl.at(off)));
        HB_LAUNCH(kKernKsTail, c, launch_mulrelin_tail_mixed(ctx_->dev(), h.tw_inv, h.logn, D, ACC, addend ? addend + off * c2 : nullptr, out2 + off * c2, c, s));
        continue;
      }
      HB_LAUNCH(kKernMulHead, c * (square ? 2 : 4), launch_mul_head(ctx_->dev(), h.tw_fwd, h.logn, true, (int)h.pack_mul | (h.conv_grid == 1 ? 4 : 0), kneed, a + off * c2, b + off * c2, ext, c, s, square ? 2u : 4u, heads, (u32)off, per));
      HB_LAUNCH(kKernMulMid, c, launch_mul_mid(ctx_->dev(), h.tw_fwd, h.tw_inv, h.logn, ctx_->dev()->mid_res_dp, h.mid_ndp, ctx_->dev()->mid_res_d, h.mid_nd, ctx_->dev()->mid_res_i, h.mid_ni, ext, D, c, s, square));
      if (fuse_head_) {
        HB_LAUNCH(kKernKsHead, c, launch_mulrelin_head(ctx_->dev(), h.tw_inv, h.tw_fwd, h.logn, (int)h.pack_mul, h.conv_grid != 0, (int)h.pack_ks, kneed, D, T, c, s));
      } else {
        HB_LAUNCH(kKernMulTail, c, launch_mul_tail(ctx_->dev(), h.tw_inv, h.logn, true, (int)h.pack_mul, h.conv_grid != 0, kneed, D, C2, c, s, 2, 1));
        HB_LAUNCH(kKernKsHead, c, launch_ks_head(ctx_->dev(), h.tw_fwd, h.logn, (int)h.pack_ks, false, K, C2, c2_words, T, c, s));
      }
      HB_LAUNCH(kKernKsMid, c, launch_ks_mid(ctx_->dev(), h.tw_fwd, h.tw_inv, h.logn, h, T, rk.key, ACC, c, s, kl.at(off)));
      HB_LAUNCH(kKernKsTail, c, launch_mulrelin_tail(ctx_->dev(), h.tw_inv, h.logn, (int)h.pack_mul, h.conv_grid != 0, (int)h.pack_ks, kneed, D, ACC,
                                                    addend ? addend + off * c2 : nullptr, out2 + off * c2, c, s, members, (u32)off, per));
    }
    return members ? (int)kOk : note_result(out2, 2, K, count, s);
  }
  const size_t chunk = chunk_ops_;
  ScratchGuard sg(pool_, std::min(chunk, count) * cs * sizeof(u64), s);
  if (!sg.p) return kOutOfMemory;
  for (size_t off = 0; off < count; off += chunk) {
    const size_t c = std::min(chunk, count - off);
    int rc = multiply(a + off * c2, 2, b + off * c2, 2, (u64*)sg.p, c, s, false);
    if (rc) return rc;
    KeySel sub = rk;  // the chunk's items are items first + off ... of the selection
    sub.first = rk.first + off;
    rc = relinearize((const u64*)sg.p, sub, out2 + off * c2, c, s, addend ? addend + off * c2 : nullptr, false);
    if (rc) return rc;
  }
  return note_result(out2, 2, h.K, count, s);
}

int Evaluator::apply_galois(const u64* ct2, u32 elt, const KeySel& key, u64* out2, size_t count, hipStream_t s, const u64* addend) {
  const DevCtx& h = ctx_->host();
  const u32 n = h.n, K = h.K;
  if (!(elt & 1) || elt >= 2 * n) return kInvalidArg;
  if (h.KK < 2 || !key.present()) return kNoKey;
  if (h.logn > 15) return kUnsupported;
  // g^{-1} mod 2n (Newton iteration, g odd)
  u64 inv = 1;
  for (int i = 0; i < 6; i++) inv = inv * (2 - (u64)elt * inv);
  const u32 ginv = (u32)(inv & (2 * n - 1));
  const size_t chunk = std::max<size_t>(1, std::min<size_t>(chunk_ops_, 65535 / ((size_t)h.KK * K)));
  const size_t rot_words = (size_t)2 * K * n;
  const size_t cc = std::min(chunk, count);
  KeyMapLease kl;
  if (int rc = stage_keymap(key, count, chunk, s, kl)) return rc;
  // r06: the split kernels read sigma_g(c1) (head) and sigma_g(c0) (tail) THROUGH the automorphism -- no rotated copy, no galois
  // launch (2 polynomials written and read again per rotation).  Not in place: the tail would gather from what it overwrites
  // (SEAL's NAF chains rotate their intermediate in place: those steps keep the copy).
  const size_t last = count % chunk ? count % chunk : cc;  // every chunk, the short last one included, must take the split kernels
  const bool fused = fuse_galois_ && ct2 != out2 && ks_split_for(cc) && ks_split_for(last);
  ScratchGuard sg(pool_, cc * ((fused ? 0 : rot_words) + ks_scratch_words()) * sizeof(u64), s);
  if (!sg.p) return kOutOfMemory;
  u64* rot = (u64*)sg.p;
  u64* ks = rot + (fused ? 0 : cc * rot_words);
  for (size_t off = 0; off < count; off += chunk) {
    const size_t c = std::min(chunk, count - off);
    if (fused) {
      const u64* in = ct2 + off * rot_words;
      int rc = key_switch(in + (size_t)K * n, rot_words, key.key, in, rot_words, 1u, out2 + off * rot_words, c, ks, s, addend ? addend + off * rot_words : nullptr,
                          kl.at(off), ginv);
      if (rc) return rc;
      continue;
    }
    HB_LAUNCH(kKernGalois, c * 2, launch_galois(ctx_->dev(), n, K, ct2 + off * rot_words, rot, c * 2, ginv, s));
    // base = (sigma(c0), 0); target = sigma(c1)
    int rc = key_switch(rot + (size_t)K * n, rot_words, key.key, rot, rot_words, 1u, out2 + off * rot_words, c, ks, s, addend ? addend + off * rot_words : nullptr,
                        kl.at(off));
    if (rc) return rc;
  }
  return note_result(out2, 2, K, count, s);
}

static int eltwise_chunks(Profiler& prof_, const DevCtx* dev, u32 n, u32 K, const u64* a, const u64* b, u64* out, size_t residue_polys, int mode,
                          hipStream_t s) {
  const size_t step = (65535 / K) * K;
  for (size_t off = 0; off < residue_polys; off += step) {
    const size_t cnt = std::min(step, residue_polys - off);
    HB_LAUNCH(kKernEltwise, cnt, launch_eltwise(dev, n, a + off * n, b ? b + off * n : nullptr, out + off * n, cnt, mode, s));
  }
  return kOk;
}

// out[op] (u64[count][size][K-1][N], the next level's layout) = mod_switch_to_next(ct[op]) : SEAL Evaluator::mod_switch_to_next
// for BFV = divide-and-round every polynomial by the last data prime
int Evaluator::mod_switch_next(const u64* ct, u32 size, u64* out, size_t count, hipStream_t s) {
  const DevCtx& h = ctx_->host();
  if (h.K < 2) return kInvalidArg;
  const size_t polys = count * size;
  for (size_t off = 0; off < polys; off += 65535) {
    const size_t c = std::min<size_t>(65535, polys - off);
    HB_CHECK(launch_mod_switch(ctx_->dev(), h.n, ct + off * (size_t)h.K * h.n, out + off * (size_t)(h.K - 1) * h.n, c, s));
  }
  return note_result(out, size, h.K - 1, count, s);
}

int Evaluator::add(const u64* a, const u64* b, u64* out, u32 size, size_t count, hipStream_t s) {
  if (size < 2) return kInvalidArg;
  if (int rc = eltwise_chunks(prof_, ctx_->dev(), ctx_->n(), ctx_->K(), a, b, out, count * size * ctx_->K(), 0, s)) return rc;
  return note_result(out, size, ctx_->K(), count, s);
}

int Evaluator::sub(const u64* a, const u64* b, u64* out, u32 size, size_t count, hipStream_t s) {
  if (size < 2) return kInvalidArg;
  if (int rc = eltwise_chunks(prof_, ctx_->dev(), ctx_->n(), ctx_->K(), a, b, out, count * size * ctx_->K(), 1, s)) return rc;
  return note_result(out, size, ctx_->K(), count, s);
}

int Evaluator::negate(const u64* a, u64* out, u32 size, size_t count, hipStream_t s) {
  if (size < 1) return kInvalidArg;
  if (int rc = eltwise_chunks(prof_, ctx_->dev(), ctx_->n(), ctx_->K(), a, nullptr, out, count * size * ctx_->K(), 2, s)) return rc;
  return note_result(out, size, ctx_->K(), count, s);
}

static int plain_addsub(Context* ctx, const u64* ct, u32 size, const u64* plain, size_t pstride, u64* out, size_t count, int sub, hipStream_t s) {
  if (size < 2) return kInvalidArg;
  const size_t cs = ctx->ct_words(size);
  if (out != ct) HB_CHECK(hipMemcpyAsync(out, ct, count * cs * sizeof(u64), hipMemcpyDeviceToDevice, s));
  for (size_t off = 0; off < count; off += 65535) {
    const size_t c = std::min<size_t>(65535, count - off);
    HB_CHECK(launch_plain_addsub(ctx->dev(), ctx->n(), out + off * cs, cs, plain + off * pstride, pstride, c, sub, s));
  }
  return kOk;
}

int Evaluator::add_plain(const u64* ct, u32 size, const u64* plain, size_t pstride, u64* out, size_t count, hipStream_t s) {
  if (int rc = plain_addsub(ctx_, ct, size, plain, pstride, out, count, 0, s)) return rc;
  return note_result(out, size, ctx_->K(), count, s);
}

int Evaluator::sub_plain(const u64* ct, u32 size, const u64* plain, size_t pstride, u64* out, size_t count, hipStream_t s) {
  if (int rc = plain_addsub(ctx_, ct, size, plain, pstride, out, count, 1, s)) return rc;
  return note_result(out, size, ctx_->K(), count, s);
}

int Evaluator::multiply_plain(const u64* ct, u32 size, const u64* plain, size_t pstride, u64* out, size_t count, hipStream_t s) {
  const DevCtx& h = ctx_->host();
  if (size < 2) return kInvalidArg;
  if (h.logn > 15) return kUnsupported;
  const u32 n = h.n, K = h.K;
  const size_t cs = ctx_->ct_words(size);
  const bool shared = pstride == 0;
  const size_t chunk = std::max<size_t>(1, std::min<size_t>(chunk_ops_, 65535 / ((size_t)size * K)));
  ScratchGuard sg(pool_, (shared ? 1 : chunk) * ((size_t)K * n * sizeof(u64) + sizeof(u32)), s);
  if (!sg.p) return kOutOfMemory;
  u64* pl = (u64*)sg.p;
  u32* nonzero = (u32*)(pl + (shared ? 1 : chunk) * (size_t)K * n);  // per-plaintext non-zero counts (monomial detection)
  std::vector<u32> mods;
  for (u32 i = 0; i < K; i++) mods.push_back(i);
  const NttPlan plan = make_plan(1, mods);
  if (shared) {
    HB_CHECK(launch_plain_lift(ctx_->dev(), n, plain, 0, pl, 1, nonzero, s));
    HB_LAUNCH(kKernNttFwd, K, launch_ntt(ctx_->dev(), h.tw_fwd, h.logn, pl, K, plan, false, 0, s));
  }
  const bool fused = h.logn >= 10 && h.logn <= 14;
  bool any_d = false, any_i = false;
  for (u32 i = 0; i < K; i++) (h.mod[i].use_f64 ? any_d : any_i) = true;
  if (out != ct && !fused) HB_CHECK(hipMemcpyAsync(out, ct, count * cs * sizeof(u64), hipMemcpyDeviceToDevice, s));
  for (size_t off = 0; off < count; off += chunk) {
    const size_t c = std::min(chunk, count - off);
    if (!shared) {
      HB_CHECK(launch_plain_lift(ctx_->dev(), n, plain + off * pstride, pstride, pl, c, nonzero, s));
      HB_LAUNCH(kKernNttFwd, c * K, launch_ntt(ctx_->dev(), h.tw_fwd, h.logn, pl, c * K, plan, false, 0, s));
    }
    if (fused) {
      // transform -> product -> inverse transform of every residue polynomial in one kernel: the ciphertext crosses HBM twice
      HB_LAUNCH(kKernPlain, c, launch_ct_plain(ctx_->dev(), h.tw_fwd, h.tw_inv, h.logn, K, any_d, any_i, pl, shared ? 0 : (size_t)K * n, ct + off * cs, out + off * cs, size, c, s));
      continue;
    }
    u64* x = out + off * cs;
    HB_LAUNCH(kKernNttFwd, c * size * K, launch_ntt(ctx_->dev(), h.tw_fwd, h.logn, x, c * size * K, plan, false, 0, s));
    HB_CHECK(launch_dyadic_plain(ctx_->dev(), n, K, x, size, pl, shared ? 0 : (size_t)K * n, c, s));
    HB_LAUNCH(kKernNttInv, c * size * K, launch_ntt(ctx_->dev(), h.tw_inv, h.logn, x, c * size * K, plan, true, 0, s));
  }
  return note_result(out, size, K, count, s);
}

int Evaluator::multiply_plain_mono(const u64* ct, u32 size, u64 coeff, u32 exponent, u64* out, size_t count, hipStream_t s) {
  const DevCtx& h = ctx_->host();
  if (size < 2 || exponent >= h.n || coeff == 0 || coeff >= h.t) return kInvalidArg;
  const u32 n = h.n, K = h.K;
  // SEAL multiply_plain_normal, monomial branch: with the fast plain lift the coefficient is used as is,
  // otherwise values >= (t+1)/2 are lifted to (coeff - t) mod q_i.
  std::vector<u64> rns(K);
  for (u32 i = 0; i < K; i++) {
    const u64 q = h.mod[i].q;
    if (h.fast_plain_lift || coeff < h.t_half_up)
      rns[i] = coeff % q;
    else {
      const u64 d = (h.t - coeff) % q;
      rns[i] = d ? q - d : 0;
    }
  }
  const size_t cs = ctx_->ct_words(size);
  const bool inplace = out == ct;
  ScratchGuard sg(pool_, (K + (inplace ? count * cs : 0)) * sizeof(u64), s);
  if (!sg.p) return kOutOfMemory;
  u64* d_rns = (u64*)sg.p;
  HB_CHECK(hipMemcpyAsync(d_rns, rns.data(), K * sizeof(u64), hipMemcpyHostToDevice, s));
  HB_CHECK(hipStreamSynchronize(s));  // rns is a host temporary
  const u64* src = ct;
  if (inplace) {
    u64* tmp = d_rns + K;
    HB_CHECK(hipMemcpyAsync(tmp, ct, count * cs * sizeof(u64), hipMemcpyDeviceToDevice, s));
    src = tmp;
  }
  const size_t total = count * size * K, step = (65535 / K) * K;
  for (size_t off = 0; off < total; off += step) {
    const size_t cnt = std::min(step, total - off);
    HB_CHECK(launch_mono_mul(ctx_->dev(), n, src + off * n, out + off * n, cnt, d_rns, exponent, s));
  }
  return note_result(out, size, K, count, s);
}

int Evaluator::nonzero_tail(const u64* ct, u32 size, u32* flags, size_t count, hipStream_t s) {
  if (size < 2) return kInvalidArg;
  const size_t cs = ctx_->ct_words(size), skip = ctx_->ct_words(1);
  for (size_t off = 0; off < count; off += 65535) {
    const size_t c = std::min<size_t>(65535, count - off);
    HB_CHECK(launch_nonzero_tail(ct + off * cs, cs, skip, flags + off, c, s));
  }
  return kOk;
}

}  // namespace hipbfv
