// sunscreen_amd/csrc/evaluator_client.cpp -- batched BatchEncoder / Decryptor / Encryptor on the device
// (the steps either side of the evaluator path; kernels in kernels_client.hip).
//
// Reference interfaces: seal_fhe/src/encoder.rs:75-190 (BatchEncoder_Encode1/2, Decode1/2),
// seal_fhe/src/encryptor_decryptor.rs:238-254 (Encryptor_Encrypt), :618-629 (Decryptor_Decrypt).
#include <algorithm>
#include <vector>

#include "evaluator.hpp"
#include "kernels.hpp"

namespace hipbfv {

namespace {
#define HC_CHECK(expr)                      \
  do {                                      \
    if ((expr) != hipSuccess) return kHipError; \
  } while (0)

#define HB_LAUNCH_CLIENT(id, units, expr)     \
  do {                                        \
    prof_.begin(id, units, s);                \
    hipError_t e__ = (expr);                  \
    prof_.end(s);                             \
    if (e__ != hipSuccess) return kHipError;  \
  } while (0)

NttPlan single_mod_plan(u32 m) {
  NttPlan pl{};
  pl.div = 1;
  pl.period = 1;
  pl.mod[0] = (unsigned char)m;
  return pl;
}
NttPlan range_plan(u32 count) {
  NttPlan pl{};
  pl.div = 1;
  pl.period = count;
  for (u32 i = 0; i < count; i++) pl.mod[i] = (unsigned char)i;
  return pl;
}
}  // namespace

// plain[op] = BatchEncoder_Encode(values[op]); values: u64[count][N] (or int64 when is_signed); *bad_host != 0 if a
// value lay outside the plain modulus (the plaintexts of such ops are not meaningful)
int Evaluator::batch_encode(const u64* values, u64* plain, size_t count, bool is_signed, u32* bad_host, hipStream_t s) {
  const DevCtx& h = ctx_->host();
  if (!ctx_->batching()) return kUnsupported;
  ScratchGuard flag(pool_, sizeof(u32), s);
  if (!flag.p) return kOutOfMemory;
  HC_CHECK(hipMemsetAsync(flag.p, 0, sizeof(u32), s));
  const NttPlan plan = single_mod_plan(h.t_mod);
  for (size_t off = 0; off < count; off += 65535) {
    const size_t c = std::min<size_t>(65535, count - off);
    HC_CHECK(launch_batch_scatter(ctx_->dev(), h.n, ctx_->batch_index_map(), values + off * h.n, plain + off * h.n, c, is_signed ? 1 : 0, (u32*)flag.p, s));
    HC_CHECK(launch_ntt(ctx_->dev(), h.tw_inv, h.logn, plain + off * h.n, c, plan, true, 0, s));
  }
  u32 bad = 0;
  HC_CHECK(hipMemcpyAsync(&bad, flag.p, sizeof(u32), hipMemcpyDeviceToHost, s));
  HC_CHECK(hipStreamSynchronize(s));
  if (bad_host) *bad_host = bad;
  return kOk;
}

int Evaluator::batch_decode(const u64* plain, u64* values, size_t count, bool is_signed, hipStream_t s) {
  const DevCtx& h = ctx_->host();
  if (!ctx_->batching()) return kUnsupported;
  const size_t chunk = std::min<size_t>(65535, std::max<size_t>(1, chunk_ops_ * 8));
  ScratchGuard tmp(pool_, std::min(chunk, count) * h.n * sizeof(u64), s);
  if (!tmp.p) return kOutOfMemory;
  const NttPlan plan = single_mod_plan(h.t_mod);
  for (size_t off = 0; off < count; off += chunk) {
    const size_t c = std::min(chunk, count - off);
    HC_CHECK(hipMemcpyAsync(tmp.p, plain + off * h.n, c * h.n * sizeof(u64), hipMemcpyDeviceToDevice, s));
    HC_CHECK(launch_ntt(ctx_->dev(), h.tw_fwd, h.logn, (u64*)tmp.p, c, plan, false, 0, s));
    HC_CHECK(launch_batch_gather(ctx_->dev(), h.n, ctx_->batch_index_map(), (const u64*)tmp.p, values + off * h.n, c, is_signed ? 1 : 0, s));
  }
  return kOk;
}

// plain[op] = Decryptor_Decrypt(ct[op]) ; ct: u64[count][size][K][N], sk: u64[KK][N] in NTT form (SEAL SecretKey data)
int Evaluator::decrypt(const u64* ct, u32 size, const u64* sk_ntt, u64* plain, size_t count, hipStream_t s) {
  const DevCtx& h = ctx_->host();
  if (size < 2 || !sk_ntt) return kInvalidArg;
  if (h.logn > 15) return kUnsupported;
  const u32 n = h.n, K = h.K;
  const size_t per = (size_t)(size - 1) * K;  // transformed residue polynomials per op
  const size_t chunk = std::max<size_t>(1, std::min<size_t>(chunk_ops_, 65535 / std::max<size_t>(per, 1)));
  const size_t cc = std::min(chunk, count);
  ScratchGuard sg(pool_, cc * (per + K) * n * sizeof(u64), s);
  if (!sg.p) return kOutOfMemory;
  u64* ctn = (u64*)sg.p;
  u64* acc = ctn + cc * per * n;
  const NttPlan plan = range_plan(K);
  const size_t cs = ctx_->ct_words(size);
  for (size_t off = 0; off < count; off += chunk) {
    const size_t c = std::min(chunk, count - off);
    // polys 1.. of every op, NTT'd: c_p (.) s^p accumulates in the transform domain
    HC_CHECK(hipMemcpy2DAsync(ctn, per * n * sizeof(u64), ct + off * cs + (size_t)K * n, cs * sizeof(u64), per * n * sizeof(u64), c,
                              hipMemcpyDeviceToDevice, s));
    HB_LAUNCH_CLIENT(kKernNttFwd, c * per, launch_ntt(ctx_->dev(), h.tw_fwd, h.logn, ctn, c * per, plan, false, 0, s));
    if (size == 2 && h.logn <= 14) {  // c1 (.) s is formed while the inverse transform loads its input
      HB_LAUNCH_CLIENT(kKernNttInv, c * K, launch_ntt_inv_dyadic(ctx_->dev(), h.tw_inv, h.logn, ctn, sk_ntt, acc, K, 1, h.KK, c, s));
    } else {
      HC_CHECK(launch_dot_secret(ctx_->dev(), n, K, ctn, size, sk_ntt, acc, c, s));
      HB_LAUNCH_CLIENT(kKernNttInv, c * K, launch_ntt(ctx_->dev(), h.tw_inv, h.logn, acc, c * K, plan, true, 0, s));
    }
    HC_CHECK(launch_decrypt_round(ctx_->dev(), n, ct + off * cs, size, acc, plain + off * n, c, s));
  }
  return kOk;
}

// ---- KeyGenerator (seal_fhe/src/key_generator.rs:20-200; SEAL keygenerator.cpp) ----
// sk_coeff / sk_ntt: u64[KK][N] -- the ternary secret in coefficient form (kept for the Galois keys) and its transform
int Evaluator::keygen_secret(const RngSeed& seed, u64* sk_coeff, u64* sk_ntt, hipStream_t s) {
  const DevCtx& h = ctx_->host();
  if (h.logn > 15) return kUnsupported;
  HC_CHECK(launch_keygen_ternary(ctx_->dev(), h.n, seed, 0, sk_coeff, s));
  HC_CHECK(hipMemcpyAsync(sk_ntt, sk_coeff, (size_t)h.KK * h.n * sizeof(u64), hipMemcpyDeviceToDevice, s));
  HC_CHECK(launch_ntt(ctx_->dev(), h.tw_fwd, h.logn, sk_ntt, h.KK, range_plan(h.KK), false, 0, s));
  return kOk;
}

// `count` key-level encryptions of zero under sk_ntt, the z-th one carrying w (.) (q_sp mod q_z) on residue z when w is
// given: count = 1, w = nullptr -> public key u64[2][KK][N]; count = K, w = s^2 or sigma_g(s) (NTT form) -> one
// key-switching key u64[K][2][KK][N].  `stream` keys the randomness (distinct per key).
int Evaluator::keygen_zero_encryptions(const RngSeed& seed, u64 stream, const u64* sk_ntt, const u64* w, u64* key, u32 count, hipStream_t s) {
  const DevCtx& h = ctx_->host();
  if (h.logn > 15) return kUnsupported;
  const u32 n = h.n, KK = h.KK;
  ScratchGuard sg(pool_, (size_t)count * 2 * KK * n * sizeof(u64), s);
  if (!sg.p) return kOutOfMemory;
  u64* a = (u64*)sg.p;
  u64* e = a + (size_t)count * KK * n;
  HC_CHECK(launch_keygen_sample(ctx_->dev(), n, seed, stream << 8, a, e, count, s));
  HC_CHECK(launch_ntt(ctx_->dev(), h.tw_fwd, h.logn, e, (size_t)count * KK, range_plan(KK), false, 0, s));
  HC_CHECK(launch_keygen_assemble(ctx_->dev(), n, KK, a, e, sk_ntt, w, key, count, s));
  return kOk;
}

// relin: w = s^2; galois element g: w = NTT(sigma_g(s)).  key: u64[K][2][KK][N]
int Evaluator::keygen_kswitch(const RngSeed& seed, u64 stream, const u64* sk_coeff, const u64* sk_ntt, u32 galois_elt, u64* key, hipStream_t s) {
  const DevCtx& h = ctx_->host();
  if (h.KK < 2) return kNoKey;  // no special prime: SEAL refuses to create key-switching keys
  const u32 n = h.n, KK = h.KK;
  ScratchGuard sg(pool_, (size_t)KK * n * sizeof(u64), s);
  if (!sg.p) return kOutOfMemory;
  u64* w = (u64*)sg.p;
  if (galois_elt == 0) {
    HC_CHECK(launch_keygen_square(ctx_->dev(), n, KK, sk_ntt, w, s));
  } else {
    if (!(galois_elt & 1) || galois_elt >= 2 * n) return kInvalidArg;
    u64 inv = 1;
    for (int i = 0; i < 6; i++) inv = inv * (2 - (u64)galois_elt * inv);
    HC_CHECK(launch_keygen_galois(ctx_->dev(), n, KK, sk_coeff, w, (u32)(inv & (2 * n - 1)), s));
    HC_CHECK(launch_ntt(ctx_->dev(), h.tw_fwd, h.logn, w, KK, range_plan(KK), false, 0, s));
  }
  return keygen_zero_encryptions(seed, stream, sk_ntt, w, key, h.K, s);
}

// ---- plaintext-matrix x ciphertext-vector products (the first loop nest of examples/pir/src/main.rs:16-45) ----
// pntt[op][K][N] = NTT(centred lift of plain[op]) -- what SEAL's multiply_plain computes internally for its plaintext
// operand (including the monomial rule, kernels.hip plain_lift_kernel).  A static database is transformed once.
int Evaluator::plain_to_ntt(const u64* plain, size_t pstride, u64* pntt, size_t count, hipStream_t s, int watch_zero) {
  const DevCtx& h = ctx_->host();
  if (h.logn > 15) return kUnsupported;
  const u32 n = h.n, K = h.K;
  const size_t chunk = std::max<size_t>(1, 65535 / K);
  ScratchGuard nz(pool_, std::min(chunk, count) * sizeof(u32), s);
  if (!nz.p) return kOutOfMemory;
  const NttPlan plan = range_plan(K);
  for (size_t off = 0; off < count; off += chunk) {
    const size_t c = std::min(chunk, count - off);
    u64* out = pntt + off * (size_t)K * n;
    HC_CHECK(launch_plain_lift(ctx_->dev(), n, plain + off * pstride, pstride, out, c, (u32*)nz.p, s));
    if (watch_zero && watch_status()) HC_CHECK(launch_zero_plain_watch((const u32*)nz.p, watch_zero == 1 ? (u32)off : 0u, watch_zero == 1 ? 1u : 0u, (u32)c, watch_status(), s));
    HB_LAUNCH_CLIENT(kKernNttFwd, c * K, launch_ntt(ctx_->dev(), h.tw_fwd, h.logn, out, c * K, plan, false, 0, s));
  }
  return kOk;
}

// ctn = NTT of every polynomial of ct (u64[count][size][K][N]); may be in place
int Evaluator::ct_to_ntt(const u64* ct, u32 size, u64* ctn, size_t count, hipStream_t s) {
  const DevCtx& h = ctx_->host();
  if (h.logn > 15) return kUnsupported;
  const u32 K = h.K;
  const size_t polys = count * size * K;
  if (ctn != ct) HC_CHECK(hipMemcpyAsync(ctn, ct, polys * h.n * sizeof(u64), hipMemcpyDeviceToDevice, s));
  const NttPlan plan = range_plan(K);
  const size_t step = (65535 / K) * K;
  for (size_t off = 0; off < polys; off += step) {
    const size_t c = std::min(step, polys - off);
    HB_LAUNCH_CLIENT(kKernNttFwd, c, launch_ntt(ctx_->dev(), h.tw_fwd, h.logn, ctn + off * h.n, c, plan, false, 0, s));
  }
  return kOk;
}

// out[row] = sum_j multiply_plain(ct_j, plain[row][j]) for size-2 ciphertexts, from the transformed operands:
// ctn: u64[cols][2][K][N] (ct_to_ntt), pntt: u64[rows][cols][K][N] (plain_to_ntt), out: u64[rows][2][K][N] coefficient form.
// Bit-identical to SEAL's sequence of multiply_plain and add (every step there is exact modular arithmetic).
int Evaluator::dot_plain_ntt(const u64* ctn, u32 cols, const u64* pntt, u32 rows, u64* out, hipStream_t s) {
  const DevCtx& h = ctx_->host();
  if (h.logn > 15) return kUnsupported;
  if (!cols || !rows) return kInvalidArg;
  const u32 n = h.n, K = h.K;
  const NttPlan plan = range_plan(K);
  const u32 step = 65535 / (2 * K) / 8 * 8;  // rows per launch: the inverse transform covers rows * 2 * K polynomials
  for (u32 off = 0; off < rows; off += step) {
    const u32 c = std::min(step, rows - off);
    u64* o = out + (size_t)off * 2 * K * n;
    HB_LAUNCH_CLIENT(kKernPlain, (size_t)c * cols, launch_dot_plain(ctx_->dev(), n, K, ctn, cols, pntt + (size_t)off * cols * K * n, c, o, s));
    HB_LAUNCH_CLIENT(kKernNttInv, (size_t)c * 2 * K, launch_ntt(ctx_->dev(), h.tw_inv, h.logn, o, (size_t)c * 2 * K, plan, true, 0, s));
  }
  return kOk;
}

// The graph executor's matrix-vector product (program_plan.cpp): as dot_plain_ntt with the plaintexts behind a descriptor table
// and a batch dimension.
int Evaluator::dot_plain_tab(const u64* ctn, u32 cols, const PlainNttRef* tab, u32 rows, u32 batch, u64* out, hipStream_t s) {
  const DevCtx& h = ctx_->host();
  if (h.logn > 15) return kUnsupported;
  if (!cols || !rows || !batch) return kInvalidArg;
  const u32 n = h.n, K = h.K;
  const NttPlan plan = range_plan(K);
  HB_LAUNCH_CLIENT(kKernPlain, (size_t)rows * cols * batch, launch_dot_plain_tab(ctx_->dev(), n, K, ctn, cols, tab, rows, batch, out, s));
  const size_t polys = (size_t)rows * batch * 2 * K, step = (65535 / K) * K;
  for (size_t off = 0; off < polys; off += step) {
    const size_t c = std::min(step, polys - off);
    HB_LAUNCH_CLIENT(kKernNttInv, c, launch_ntt(ctx_->dev(), h.tw_inv, h.logn, out + off * n, c, plan, true, 0, s));
  }
  return kOk;
}

int Evaluator::multiply_plain_ntt(const u64* ct, u32 size, const u64* pntt, size_t pnstride, u64* out, size_t count, hipStream_t s) {
  const DevCtx& h = ctx_->host();
  if (size < 2 || !pntt) return kInvalidArg;
  if (h.logn > 15) return kUnsupported;
  const u32 n = h.n, K = h.K;
  if (int rc = ct_to_ntt(ct, size, out, count, s)) return rc;
  const NttPlan plan = range_plan(K);
  const size_t cs = ctx_->ct_words(size);
  for (size_t off = 0; off < count; off += 65535) {
    const size_t c = std::min<size_t>(65535, count - off);
    HC_CHECK(launch_dyadic_plain(ctx_->dev(), n, K, out + off * cs, size, pntt + off * pnstride, pnstride, c, s));
  }
  const size_t polys = count * size * K, step = (65535 / K) * K;
  for (size_t off = 0; off < polys; off += step) {
    const size_t c = std::min(step, polys - off);
    HB_LAUNCH_CLIENT(kKernNttInv, c, launch_ntt(ctx_->dev(), h.tw_inv, h.logn, out + off * n, c, plan, true, 0, s));
  }
  return note_result(out, size, K, count, s);
}

int Evaluator::note_nary(const NaryOut* douts, u32 nouts, u32 batch, hipStream_t s) {
  u32* status = watch_status();
  if (!status || !nouts) return kOk;
  HC_CHECK(launch_transparent_watch_nary(ctx_->dev(), douts, nouts, batch, status, s));
  return kOk;
}

// phase[op][i] = (c0 + c1*s + c2*s^2 ...) mod q_i in coefficient form: u64[count][K][N].  The quantity SEAL's
// Decryptor::invariant_noise_budget measures (diagnostics; not a hot path: one small launch per op for the last step)
int Evaluator::phase(const u64* ct, u32 size, const u64* sk_ntt, u64* out, size_t count, hipStream_t s) {
  const DevCtx& h = ctx_->host();
  if (size < 2 || !sk_ntt) return kInvalidArg;
  if (h.logn > 15) return kUnsupported;
  const u32 n = h.n, K = h.K;
  const size_t per = (size_t)(size - 1) * K;
  ScratchGuard sg(pool_, per * n * sizeof(u64), s);
  if (!sg.p) return kOutOfMemory;
  u64* ctn = (u64*)sg.p;
  const NttPlan plan = range_plan(K);
  const size_t cs = ctx_->ct_words(size);
  for (size_t op = 0; op < count; op++) {
    const u64* c = ct + op * cs;
    u64* acc = out + op * (size_t)K * n;
    HC_CHECK(hipMemcpyAsync(ctn, c + (size_t)K * n, per * n * sizeof(u64), hipMemcpyDeviceToDevice, s));
    HC_CHECK(launch_ntt(ctx_->dev(), h.tw_fwd, h.logn, ctn, per, plan, false, 0, s));
    HC_CHECK(launch_dot_secret(ctx_->dev(), n, K, ctn, size, sk_ntt, acc, 1, s));
    HC_CHECK(launch_ntt(ctx_->dev(), h.tw_inv, h.logn, acc, K, plan, true, 0, s));
    HC_CHECK(launch_eltwise(ctx_->dev(), n, acc, c, acc, K, 0, s));
  }
  return kOk;
}

// ct2[op] = Encryptor_Encrypt(plain[op]) under the public key pk: u64[2][KK][N] (NTT form, key level).
// Randomness: ChaCha20 blocks keyed by `seed.secret` (rng.hpp), counter = (coefficient, first_op + op): reproducible and independent of
// the chunking.  plain: u64[count][N] (pstride = N) or one shared plaintext (pstride = 0).
int Evaluator::encrypt(const u64* plain, size_t pstride, const u64* pk, const RngSeed& seed, u64 first_op, u64* ct2, size_t count, hipStream_t s) {
  const DevCtx& h = ctx_->host();
  if (!pk) return kInvalidArg;
  if (h.logn > 15) return kUnsupported;
  const u32 n = h.n, K = h.K, KK = h.KK;
  const size_t chunk = std::max<size_t>(1, std::min<size_t>(chunk_ops_, 65535 / (2 * (size_t)KK)));
  const size_t cc = std::min(chunk, count);
  // u[KK] + c[2][KK] residue polynomials per op; the error polynomials are regenerated where they are consumed
  ScratchGuard sg(pool_, cc * 3 * (size_t)KK * n * sizeof(u64), s);
  if (!sg.p) return kOutOfMemory;
  u64* u = (u64*)sg.p;
  u64* c2 = u + cc * (size_t)KK * n;
  const NttPlan plan = range_plan(KK);
  for (size_t off = 0; off < count; off += chunk) {
    const size_t c = std::min(chunk, count - off);
    HC_CHECK(launch_encrypt_sample(ctx_->dev(), n, seed, first_op + off, u, nullptr, c, s));
    HB_LAUNCH_CLIENT(kKernNttFwd, c * KK, launch_ntt(ctx_->dev(), h.tw_fwd, h.logn, u, c * KK, plan, false, 0, s));
    if (h.logn <= 14) {  // pk (.) NTT(u) is formed while the inverse transform loads its input
      HB_LAUNCH_CLIENT(kKernNttInv, c * 2 * KK, launch_ntt_inv_dyadic(ctx_->dev(), h.tw_inv, h.logn, u, pk, c2, KK, 2, KK, c, s));
    } else {
      HC_CHECK(launch_encrypt_dyadic(ctx_->dev(), n, KK, u, pk, c2, c, s));
      HB_LAUNCH_CLIENT(kKernNttInv, c * 2 * KK, launch_ntt(ctx_->dev(), h.tw_inv, h.logn, c2, c * 2 * KK, plan, true, 0, s));
    }
    // + e, SEAL's divide-and-round by the special prime (the mod_switch of the fresh key-level encryption) and
    // + round(q/t * m) with SEAL's rounding correction (multiply_add_plain_with_scaling_variant), in one pass
    HC_CHECK(launch_encrypt_finish(ctx_->dev(), n, seed, first_op + off, c2, plain + off * pstride, pstride, ct2 + off * 2 * K * n, c, s));
  }
  return kOk;
}

// First K residue rows of each of `polys` key-level polynomials: src u64[polys][KK][N] -> dst u64[polys][K][N]
static hipError_t take_data_rows(const DevCtx& h, const u64* src, u64* dst, u32 polys, hipStream_t s) {
  const size_t row = (size_t)h.n * sizeof(u64);
  return hipMemcpy2DAsync(dst, h.K * row, src, h.KK * row, h.K * row, polys, hipMemcpyDeviceToDevice, s);
}

// One public-key encryption that also hands back what was sampled (the fork's Encryptor_EncryptReturnComponents,
// seal_fhe/src/encryptor_decryptor.rs:268-300): u_out u64[K][N] (ternary), e_out u64[2][K][N], coefficient-form data-level
// residues.  no_special: compute (pk*u + e) on the data primes only, with no division by the special prime, so that
// c0 = floor(q/t)*m + r + pk0*u + e0 and c1 = pk1*u + e1 hold EXACTLY (what logproof proves, bfv_statement.rs:159).
int Evaluator::encrypt_components(const u64* plain, const u64* pk, const RngSeed& seed, u64 op, bool no_special, u64* ct2, u64* u_out, u64* e_out,
                                  hipStream_t s) {
  const DevCtx& h = ctx_->host();
  if (!pk) return kInvalidArg;
  if (h.logn > 15) return kUnsupported;
  const u32 n = h.n, KK = h.KK;
  ScratchGuard sg(pool_, 5 * (size_t)KK * n * sizeof(u64), s);
  if (!sg.p) return kOutOfMemory;
  u64* u = (u64*)sg.p;
  u64* e = u + (size_t)KK * n;
  u64* c2 = e + 2 * (size_t)KK * n;
  const NttPlan plan = range_plan(KK);
  HC_CHECK(launch_encrypt_sample(ctx_->dev(), n, seed, op, u, e, 1, s));
  if (u_out) HC_CHECK(take_data_rows(h, u, u_out, 1, s));
  if (e_out) HC_CHECK(take_data_rows(h, e, e_out, 2, s));
  HC_CHECK(launch_ntt(ctx_->dev(), h.tw_fwd, h.logn, u, KK, plan, false, 0, s));
  HC_CHECK(launch_encrypt_dyadic(ctx_->dev(), n, KK, u, pk, c2, 1, s));
  HC_CHECK(launch_ntt(ctx_->dev(), h.tw_inv, h.logn, c2, 2 * KK, plan, true, 0, s));
  HC_CHECK(launch_add_key_level(ctx_->dev(), n, c2, e, 2 * KK, s));
  if (KK > 1 && !no_special) {
    HC_CHECK(launch_ks_moddown(ctx_->dev(), n, c2, nullptr, 0, 0u, nullptr, ct2, 1, s));
  } else {
    HC_CHECK(take_data_rows(h, c2, ct2, 2, s));
  }
  return add_plain(ct2, 2, plain, 0, ct2, 1, s);
}

// Secret-key encryption (SEAL encrypt_zero_symmetric at the data level + scaled plaintext): c1 = a uniform,
// c0 = floor(q/t)*m + r - (a*s + e).  e_out (optional): u64[K][N] coefficient-form residues of e.
int Evaluator::encrypt_symmetric(const u64* plain, const u64* sk_ntt, const RngSeed& seed, u64 stream, u64* ct2, u64* e_out, hipStream_t s) {
  const DevCtx& h = ctx_->host();
  if (!sk_ntt) return kInvalidArg;
  if (h.logn > 15) return kUnsupported;
  const u32 n = h.n, KK = h.KK;
  ScratchGuard sg(pool_, 4 * (size_t)KK * n * sizeof(u64), s);
  if (!sg.p) return kOutOfMemory;
  u64* a = (u64*)sg.p;
  u64* e = a + (size_t)KK * n;
  u64* c = e + (size_t)KK * n;
  const NttPlan plan = range_plan(KK);
  HC_CHECK(launch_keygen_sample(ctx_->dev(), n, seed, stream << 8, a, e, 1, s));
  if (e_out) HC_CHECK(take_data_rows(h, e, e_out, 1, s));
  HC_CHECK(launch_ntt(ctx_->dev(), h.tw_fwd, h.logn, e, KK, plan, false, 0, s));
  HC_CHECK(launch_keygen_assemble(ctx_->dev(), n, KK, a, e, sk_ntt, nullptr, c, 1, s));
  HC_CHECK(launch_ntt(ctx_->dev(), h.tw_inv, h.logn, c, 2 * KK, plan, true, 0, s));
  HC_CHECK(take_data_rows(h, c, ct2, 2, s));
  return add_plain(ct2, 2, plain, 0, ct2, 1, s);
}

// Key-level NTT-form polynomials (public / secret key) as data-level coefficient-form residues: u64[polys][K][N]
int Evaluator::key_to_coeff(const u64* key, u32 polys, u64* out, hipStream_t s) {
  const DevCtx& h = ctx_->host();
  if (h.logn > 15) return kUnsupported;
  const u32 n = h.n, KK = h.KK;
  ScratchGuard sg(pool_, (size_t)polys * KK * n * sizeof(u64), s);
  if (!sg.p) return kOutOfMemory;
  u64* tmp = (u64*)sg.p;
  HC_CHECK(hipMemcpyAsync(tmp, key, (size_t)polys * KK * n * sizeof(u64), hipMemcpyDeviceToDevice, s));
  HC_CHECK(launch_ntt(ctx_->dev(), h.tw_inv, h.logn, tmp, (size_t)polys * KK, range_plan(KK), true, 0, s));
  HC_CHECK(take_data_rows(h, tmp, out, polys, s));
  return kOk;
}

int Evaluator::crt_compose(const u64* consts, u32 kc, const u64* in, u64* out, u32 polys, hipStream_t s) {
  HC_CHECK(launch_crt_compose(ctx_->dev(), ctx_->host().n, kc, consts, in, out, polys, s));
  return kOk;
}
int Evaluator::crt_decompose(u32 kc, const u64* in, u64* out, u32 polys, hipStream_t s) {
  HC_CHECK(launch_crt_decompose(ctx_->dev(), ctx_->host().n, kc, in, out, polys, s));
  return kOk;
}

}  // namespace hipbfv
