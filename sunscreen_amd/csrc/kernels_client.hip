// sunscreen_amd/csrc/kernels_client.hip -- the steps either side of the evaluator, batched:
//   BatchEncoder_Encode / Decode        (seal_fhe/src/encoder.rs:75-190)        -> slot permutation + NTT over Z_t
//   Decryptor_Decrypt                   (seal_fhe/src/encryptor_decryptor.rs:618-629) -> <ct, (1, s, s^2)> then
//                                        SEAL RNSTool::decrypt_scale_and_round in the base {t, gamma}
//   Encryptor_Encrypt (public key)      (seal_fhe/src/encryptor_decryptor.rs:238-254) -> u, e0, e1 sampling,
//                                        (pk0*u + e0, pk1*u + e1) at key level, divide-and-round by the special prime
// The transforms themselves are the library's NTT kernels (launch_ntt); this file holds the coefficient-parallel
// kernels around them.  Coefficient-parallel layout as in kernels.hip: thread = coefficient, residues N apart.
#include <hip/hip_runtime.h>
#include <algorithm>

#include "devarith.hpp"
#include "kernels.hpp"

namespace hipbfv {

namespace {
constexpr int kClientThreads = 256;
inline dim3 cgrid(u32 n, u32 y, u32 z = 1) { return dim3((n + kClientThreads - 1) / kClientThreads, y, z); }
}  // namespace

// ---- BatchEncoder ----
// plain[op][map[i]] = values[op][i] (mod t); is_signed: values are int64 in [-(t>>1), t>>1].  *bad |= 1 on a
// value outside the plain modulus (SEAL throws std::invalid_argument).
__global__ __launch_bounds__(kClientThreads) void batch_scatter_kernel(const DevCtx* __restrict__ ctx, const u32* __restrict__ map,
                                                                       const u64* __restrict__ values, u64* __restrict__ plain, int is_signed,
                                                                       u32* __restrict__ bad) {
  const u32 n = ctx->n;
  const u32 i = blockIdx.x * kClientThreads + threadIdx.x;
  const u32 op = blockIdx.y;
  if (i >= n) return;
  const u64 t = ctx->t;
  u64 v = values[(size_t)op * n + i];
  if (is_signed) {
    const long long sv = (long long)v;
    const u64 mag = sv < 0 ? (u64)(-sv) : (u64)sv;
    if (mag > (t >> 1)) {
      atomicOr(bad, 1u);
      v = 0;
    } else {
      v = sv < 0 ? t - mag : mag;
    }
  } else if (v >= t) {
    atomicOr(bad, 1u);
    v = 0;
  }
  plain[(size_t)op * n + map[i]] = v;
}

// values[op][i] = tmp[op][map[i]]; is_signed: representatives in (-(t>>1)-1, t>>1] as SEAL's Decode2
__global__ __launch_bounds__(kClientThreads) void batch_gather_kernel(const DevCtx* __restrict__ ctx, const u32* __restrict__ map,
                                                                      const u64* __restrict__ tmp, u64* __restrict__ values, int is_signed) {
  const u32 n = ctx->n;
  const u32 i = blockIdx.x * kClientThreads + threadIdx.x;
  const u32 op = blockIdx.y;
  if (i >= n) return;
  const u64 t = ctx->t;
  u64 v = tmp[(size_t)op * n + map[i]];
  if (is_signed && v > (t >> 1)) v = (u64)((long long)v - (long long)t);
  values[(size_t)op * n + i] = v;
}

// ---- Decryptor ----
// acc[op][i] = sum_{p=1}^{size-1} ctn[op][p-1][i] (.) s_i^p   (everything in NTT form; s: u64[KK][N])
__global__ __launch_bounds__(kClientThreads) void dot_secret_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ ctn, u32 size,
                                                                    const u64* __restrict__ sk, u64* __restrict__ acc) {
  const u32 n = ctx->n, K = ctx->K;
  const u32 x = blockIdx.x * kClientThreads + threadIdx.x;
  const u32 i = blockIdx.y, op = blockIdx.z;
  if (x >= n) return;
  const DevMod& dm = ctx->mod[i];
  const u64 s1 = sk[(size_t)i * n + x];
  u64 sp = s1, a = 0;
  for (u32 p = 1; p < size; p++) {
    const u64 c = ctn[(((size_t)op * (size - 1) + (p - 1)) * K + i) * n + x];
    a = add_mod(a, mul_mod(c, sp, dm), dm.q);
    sp = mul_mod(sp, s1, dm);
  }
  acc[((size_t)op * K + i) * n + x] = a;
}

// plain[op][x] = round(t * (c0 + acc) / q) mod t, SEAL RNSTool::decrypt_scale_and_round:
//   y_i = phase_i * t * gamma * (q/q_i)^{-1} mod q_i ; fast base conversion q -> {t, gamma} ; times -q^{-1} ;
//   centred gamma correction ; times gamma^{-1} mod t.
__global__ __launch_bounds__(kClientThreads) void decrypt_round_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ ct, u32 size,
                                                                       const u64* __restrict__ acc, u64* __restrict__ plain) {
  const u32 n = ctx->n, K = ctx->K;
  const u32 x = blockIdx.x * kClientThreads + threadIdx.x;
  const u32 op = blockIdx.y;
  if (x >= n) return;
  u128 at = 0, ag = 0;
  for (u32 i = 0; i < K; i++) {
    const DevMod& dm = ctx->mod[i];
    const u64 c0 = ct[((size_t)op * size * K + i) * n + x];
    const u64 phase = add_mod(c0, acc[((size_t)op * K + i) * n + x], dm.q);
    const u64 y = mul_shoup(phase, ctx->dec_scale_q[i], dm.q);
    at += (u128)y * ctx->q_to_t[i];
    ag += (u128)y * ctx->q_to_gamma[i];
  }
  const u64 t = ctx->t, gamma = ctx->gamma.q;
  const u64 a = mul_shoup(reduce128(at, ctx->tm), ctx->neg_inv_q_mod_t, t);
  const u64 g = mul_shoup(reduce128_fast(ag, ctx->gamma), ctx->neg_inv_q_mod_gamma, gamma);
  u64 r;
  if (g > (gamma >> 1))
    r = add_mod(a, reduce64(gamma - g, ctx->tm), t);
  else
    r = sub_mod(a, reduce64(g, ctx->tm), t);
  if (r) r = mul_shoup(r, ctx->inv_gamma_mod_t, t);
  plain[(size_t)op * n + x] = r;
}

// ---- Encryptor ----
// Randomness: ChaCha20 as a counter-based PRF (rng.hpp): 16 words = block(key, (coefficient, op_lo, op_hi, domain)).
// Secret material (u, e, ternary secrets) is generated under RngSeed::secret, published uniform polynomials under ::pub.
// rounded Gaussian, sigma = 3.2, clipped to |x| <= 19 (the reference builds SEAL with SEAL_USE_GAUSSIAN_NOISE=ON,
// seal_fhe/build.rs:50; the sampler is pinned statistically only -- SURVEY 8f row 3)
__device__ __forceinline__ int gauss_noise(u32 a, u32 b) {
  const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777217.0f);
  const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
  float z = sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2) * 3.2f;
  z = fminf(fmaxf(z, -19.0f), 19.0f);
  return (int)rintf(z);
}

__device__ __forceinline__ u64 small_to_residue(int v, u64 q) { return v < 0 ? q - (u64)(-v) : (u64)v; }

// u[op][KK][N] = ternary polynomial in every key-level residue; e[op][2][KK][N] = the two error polynomials
__global__ __launch_bounds__(kClientThreads) void encrypt_sample_kernel(const DevCtx* __restrict__ ctx, RngKey key, u64 op0, u64* __restrict__ u,
                                                                        u64* __restrict__ e) {
  const u32 n = ctx->n, KK = ctx->KK;
  const u32 x = blockIdx.x * kClientThreads + threadIdx.x;
  const u32 op = blockIdx.y;
  if (x >= n) return;
  const u64 gop = op0 + op;
  u32 r[5];
  chacha20_block<5>(key, x, (u32)gop, (u32)(gop >> 32), 0u, r);
  const int tern = (int)(((u64)r[0] * 3u) >> 32) - 1;
  const int e0 = gauss_noise(r[1], r[2]), e1 = gauss_noise(r[3], r[4]);
  for (u32 i = 0; i < KK; i++) {
    const u64 q = ctx->mod[i].q;
    u[((size_t)op * KK + i) * n + x] = small_to_residue(tern, q);
    if (e) {  // nullptr: the caller regenerates the errors where it consumes them (encrypt_finish_kernel)
      e[(((size_t)op * 2 + 0) * KK + i) * n + x] = small_to_residue(e0, q);
      e[(((size_t)op * 2 + 1) * KK + i) * n + x] = small_to_residue(e1, q);
    }
  }
}

// The rest of a public-key encryption in one pass over the key-level product c2 = INTT(pk (.) NTT(u)), u64[op][2][KK][N]:
// + e (regenerated from the same ChaCha block as encrypt_sample_kernel, never stored), SEAL's divide-and-round by the
// special prime (RNSTool::divide_and_round_q_last, as ks_moddown_kernel), and on polynomial 0 the scaled plaintext
// floor(q/t) m + r (plain_addsub_kernel).  plain: u64[ops or 1][N]; out: u64[op][2][K][N].
__global__ __launch_bounds__(kClientThreads) void encrypt_finish_kernel(const DevCtx* __restrict__ ctx, RngKey key, u64 op0, const u64* __restrict__ c2,
                                                                        const u64* __restrict__ plain, size_t pstride, u64* __restrict__ out) {
  const u32 n = ctx->n, K = ctx->K, KK = ctx->KK;
  const u32 x = blockIdx.x * kClientThreads + threadIdx.x;
  const u32 op = blockIdx.y;
  if (x >= n) return;
  const u64 gop = op0 + op;
  u32 r[5];
  chacha20_block<5>(key, x, (u32)gop, (u32)(gop >> 32), 0u, r);  // one block serves both polynomials' errors
  const u64 m = plain[(size_t)op * pstride + x];
  const u64 fix = (u64)(((u128)m * ctx->q_mod_t + ctx->t_half_up) / ctx->t);
  const DevMod& sp = ctx->mod[KK - 1];
#pragma unroll
  for (u32 c = 0; c < 2; c++) {
    const int err = c == 0 ? gauss_noise(r[1], r[2]) : gauss_noise(r[3], r[4]);
    const u64* acc = c2 + ((size_t)op * 2 + c) * KK * n + x;
    u64 tl = 0;
    if (KK > 1) tl = add_mod(add_mod(acc[(size_t)(KK - 1) * n], small_to_residue(err, sp.q), sp.q), ctx->qsp_half, sp.q);
    for (u32 J = 0; J < K; J++) {
      const DevMod& mj = ctx->mod[J];
      u64 d = add_mod(acc[(size_t)J * n], small_to_residue(err, mj.q), mj.q);
      if (KK > 1) {
        u64 tk = sp.q > mj.q ? reduce64(tl, mj) : tl;
        tk = sub_mod(tk, ctx->qsp_half_mod_q[J], mj.q);
        d = mul_shoup(sub_mod(d, tk, mj.q), ctx->inv_qsp_mod_q[J], mj.q);
      }
      if (c == 0) d = add_mod(d, reduce128((u128)m * ctx->q_div_t_mod_q[J] + fix, mj), mj.q);
      out[(((size_t)op * 2 + c) * K + J) * n + x] = d;
    }
  }
}

// c[op][j][i] = un[op][i] (.) pk[j][i]   (NTT form, key level)
__global__ __launch_bounds__(kClientThreads) void encrypt_dyadic_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ un,
                                                                        const u64* __restrict__ pk, u64* __restrict__ c) {
  const u32 n = ctx->n, KK = ctx->KK;
  const u32 x = blockIdx.x * kClientThreads + threadIdx.x;
  const u32 i = blockIdx.y, op = blockIdx.z;
  if (x >= n) return;
  const DevMod& dm = ctx->mod[i];
  const u64 a = un[((size_t)op * KK + i) * n + x];
#pragma unroll
  for (int j = 0; j < 2; j++)
    c[(((size_t)op * 2 + j) * KK + i) * n + x] = mul_mod(a, pk[((size_t)j * KK + i) * n + x], dm);
}

// c += e over [polys][KK][N] (key level residues)
__global__ __launch_bounds__(kClientThreads) void add_key_level_kernel(const DevCtx* __restrict__ ctx, u64* __restrict__ c, const u64* __restrict__ e) {
  const u32 n = ctx->n, KK = ctx->KK;
  const u32 x = blockIdx.x * kClientThreads + threadIdx.x;
  const u32 res = blockIdx.y;
  if (x >= n) return;
  const size_t off = (size_t)res * n + x;
  c[off] = add_mod(c[off], e[off], ctx->mod[res % KK].q);
}

// ---- KeyGenerator (seal_fhe/src/key_generator.rs:20-200) ----
// s[i][x] = ternary secret polynomial (coefficient form) in every key-level residue
__global__ __launch_bounds__(kClientThreads) void keygen_ternary_kernel(const DevCtx* __restrict__ ctx, RngKey key, u64 stream, u64* __restrict__ s) {
  const u32 n = ctx->n, KK = ctx->KK;
  const u32 x = blockIdx.x * kClientThreads + threadIdx.x;
  if (x >= n) return;
  u32 r[1];
  chacha20_block<1>(key, x, (u32)stream, (u32)(stream >> 32), 0x5Eu, r);
  const int tern = (int)(((u64)r[0] * 3u) >> 32) - 1;
  for (u32 i = 0; i < KK; i++) s[(size_t)i * n + x] = small_to_residue(tern, ctx->mod[i].q);
}

// For `count` independent "encryptions of zero" at the key level (SEAL encrypt_zero_symmetric, NTT form):
//   a[z][i][x] = uniform residue mod q_i (sampled directly in the transform domain),  e[z][i][x] = one rounded Gaussian
//   per coefficient in every residue (coefficient form; the caller transforms it).
__global__ __launch_bounds__(kClientThreads) void keygen_sample_kernel(const DevCtx* __restrict__ ctx, RngSeed seed, u64 stream0, u64* __restrict__ a,
                                                                       u64* __restrict__ e) {
  const u32 n = ctx->n, KK = ctx->KK;
  const u32 x = blockIdx.x * kClientThreads + threadIdx.x;
  const u32 z = blockIdx.y;
  if (x >= n) return;
  const u64 stream = stream0 + z;
  u32 g[2];
  chacha20_block<2>(seed.secret, x, (u32)stream, (u32)(stream >> 32), 0xE0u, g);  // the error is secret
  const int err = gauss_noise(g[0], g[1]);
  u32 blk[16];
  for (u32 i = 0; i < KK; i++) {
    const DevMod& dm = ctx->mod[i];
    // the uniform polynomial is published: its own key; one block serves four residues (4 words each)
    if ((i & 3) == 0) chacha20_block<16>(seed.pub, x, (u32)stream, (u32)(stream >> 32), 0xA0u + (i >> 2), blk);
    const u32* r = blk + 4 * (i & 3);
    // 128 uniform bits reduced mod q_i: bias below 2^-66 (SEAL rejects instead; indistinguishable at this size)
    const u128 wide = ((u128)(((u64)r[0] << 32) | r[1]) << 64) | (((u64)r[2] << 32) | r[3]);
    a[((size_t)z * KK + i) * n + x] = reduce128(wide >> 1, dm);
    e[((size_t)z * KK + i) * n + x] = small_to_residue(err, dm.q);
  }
}

// key[z] = (c0, c1) with c1 = a, c0 = -(a (.) s + e) [+ w (.) factor on residue z, when w != nullptr]   (all NTT form)
//   z = digit index of a key-switching key (factor = q_sp mod q_z, SEAL KeyGenerator::generate_one_kswitch_key) or 0 for a
//   public key (w = nullptr).  key: u64[count][2][KK][N]
__global__ __launch_bounds__(kClientThreads) void keygen_assemble_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ a,
                                                                         const u64* __restrict__ e, const u64* __restrict__ s,
                                                                         const u64* __restrict__ w, u64* __restrict__ key) {
  const u32 n = ctx->n, KK = ctx->KK;
  const u32 x = blockIdx.x * kClientThreads + threadIdx.x;
  const u32 i = blockIdx.y, z = blockIdx.z;
  if (x >= n) return;
  const DevMod& dm = ctx->mod[i];
  const u64 av = a[((size_t)z * KK + i) * n + x];
  u64 c0 = neg_mod(add_mod(mul_mod(av, s[(size_t)i * n + x], dm), e[((size_t)z * KK + i) * n + x], dm.q), dm.q);
  if (w && i == z) {
    const u64 factor = reduce64(ctx->mod[KK - 1].q, dm);
    c0 = add_mod(c0, mul_mod(w[(size_t)i * n + x], factor, dm), dm.q);
  }
  key[(((size_t)z * 2 + 0) * KK + i) * n + x] = c0;
  key[(((size_t)z * 2 + 1) * KK + i) * n + x] = av;
}

// out[i] = in[i] (.) in[i] per residue (s^2 in the transform domain), key level
__global__ __launch_bounds__(kClientThreads) void keygen_square_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ in, u64* __restrict__ out) {
  const u32 n = ctx->n;
  const u32 x = blockIdx.x * kClientThreads + threadIdx.x;
  const u32 i = blockIdx.y;
  if (x >= n) return;
  const u64 v = in[(size_t)i * n + x];
  out[(size_t)i * n + x] = mul_mod(v, v, ctx->mod[i]);
}

// out[i][o] = +-in[i][o * g^{-1} mod 2N]: the Galois automorphism of a key-level polynomial in coefficient form
__global__ __launch_bounds__(kClientThreads) void keygen_galois_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ in, u64* __restrict__ out,
                                                                       u32 ginv) {
  const u32 n = ctx->n;
  const u32 o = blockIdx.x * kClientThreads + threadIdx.x;
  const u32 i = blockIdx.y;
  if (o >= n) return;
  const u32 kk = (u32)(((u64)o * ginv) & (2 * n - 1));
  const u64 v = in[(size_t)i * n + (kk & (n - 1))];
  out[(size_t)i * n + o] = kk >= n ? neg_mod(v, ctx->mod[i].q) : v;
}

// ---- PolynomialArray (seal_fhe/src/data_structures.rs:130-304): RNS <-> multiprecision ("[poly][coeff][limb]") ----
// consts: u64 inv_punct[KC] | punct[KC][KC] (little-endian limbs of q/q_i) | q[KC] (limbs of q)
// x = sum_i [x_i * (q/q_i)^-1 mod q_i] * (q/q_i)  mod q, the canonical representative in [0, q)
template <int KC>
__global__ __launch_bounds__(kClientThreads) void crt_compose_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ consts,
                                                                     const u64* __restrict__ in, u64* __restrict__ out) {
  const u32 n = ctx->n;
  const u32 x = blockIdx.x * kClientThreads + threadIdx.x;
  const u32 p = blockIdx.y;
  if (x >= n) return;
  const u64* inv_punct = consts;
  const u64* punct = consts + KC;
  const u64* qw = consts + KC + KC * KC;
  u64 acc[KC + 1];
#pragma unroll
  for (int l = 0; l <= KC; l++) acc[l] = 0;
#pragma unroll
  for (int i = 0; i < KC; i++) {
    const u64 y = mul_mod(in[((size_t)p * KC + i) * n + x], inv_punct[i], ctx->mod[i]);
    u64 carry = 0;
#pragma unroll
    for (int l = 0; l < KC; l++) {
      const u128 t = (u128)y * punct[i * KC + l] + acc[l] + carry;
      acc[l] = (u64)t;
      carry = (u64)(t >> 64);
    }
    acc[KC] += carry;
  }
  // acc < KC * q: at most KC - 1 subtractions
  for (int round = 0; round < KC; round++) {
    bool ge = acc[KC] != 0;
    if (!ge) {
      ge = true;  // equal counts as >= (x == q -> 0)
#pragma unroll
      for (int l = KC - 1; l >= 0; l--) {
        if (acc[l] != qw[l]) {
          ge = acc[l] > qw[l];
          break;
        }
      }
    }
    if (!ge) break;
    u64 borrow = 0;
#pragma unroll
    for (int l = 0; l < KC; l++) {
      const u128 d = (u128)acc[l] - qw[l] - borrow;
      acc[l] = (u64)d;
      borrow = (u64)(d >> 64) & 1;
    }
    acc[KC] -= borrow;
  }
#pragma unroll
  for (int l = 0; l < KC; l++) out[((size_t)p * n + x) * KC + l] = acc[l];
}

// out[p][i][x] = (multiprecision in[p][x][0..KC)) mod q_i, by Horner over the limbs
__global__ __launch_bounds__(kClientThreads) void crt_decompose_kernel(const DevCtx* __restrict__ ctx, u32 KC, const u64* __restrict__ in,
                                                                       u64* __restrict__ out) {
  const u32 n = ctx->n;
  const u32 x = blockIdx.x * kClientThreads + threadIdx.x;
  const u32 i = blockIdx.y, p = blockIdx.z;
  if (x >= n) return;
  const DevMod& dm = ctx->mod[i];
  const u64* limbs = in + ((size_t)p * n + x) * KC;
  u64 r = 0;
  for (int l = (int)KC - 1; l >= 0; l--) r = reduce128(((u128)r << 64) | limbs[l], dm);
  out[((size_t)p * KC + i) * n + x] = r;
}

// ---- plaintext-matrix x ciphertext-vector product in the transform domain (examples/pir/src/main.rs:16-45) ----
// acc[row][p][i][x] = sum_j ctn[j][p][i][x] * pntt[row][j][i][x] mod q_i     (all NTT form, canonical residues)
// One thread = one coefficient of one residue, RT consecutive rows and both ciphertext polynomials: every ciphertext
// word is read once per RT rows, the plaintext matrix (the database) streams through exactly once.


// The same with TWO adjacent coefficients per thread: every access is 16 bytes per lane (1 KB contiguous per wavefront
// instruction instead of 512 B), half the memory instructions for the same bytes.  RT rows x 2 polynomials x 2 coefficients of
// 128-bit accumulators per thread.
//
// r04: the column loop is software-pipelined by hand.  The r02-r03 form tested `r0 + r < rows` inside the loop; the compiler
// turned each row into its own basic block -- load, s_waitcnt vmcnt(0), multiply -- so ONE 16-byte load per lane was in flight at
// a time and the database streamed on occupancy alone (4.4-4.9 TB/s).  Now a block of JU columns is fetched as a unit -- the RT x
// JU database words and the 2 x JU query words issued back to back -- and the NEXT block is requested before the current one is
// multiplied; rows beyond the matrix are clamped to its last row (read twice, never stored) so the loop body has no branches.
// Src: where database word (row r of the thread's RT, column j) lives: a dense matrix (dot_plain2_kernel) or a descriptor
// table (dot_plain_tab_kernel, the graph executor's form).
typedef unsigned long long pir_u64x2 __attribute__((ext_vector_type(2)));
template <int RT, int JU>
struct PirBlock {
  pir_u64x2 c0[JU], c1[JU], pv[RT][JU];
};
template <int RT, int JU, class Query, class Src>
__device__ __forceinline__ void pir_fetch(PirBlock<RT, JU>& blk, u32 j0, u32 cols, const Query& query, const Src& src) {
  // first every address (the descriptor-table form reads RT x JU wave-uniform table entries: scalar loads, requested together),
  // then every vector load back to back
  const u64* at[RT][JU];
  u32 jj[JU];
#pragma unroll
  for (int u = 0; u < JU; u++) {
    jj[u] = j0 + (u32)u < cols ? j0 + (u32)u : cols - 1;  // a partial last block re-reads the last column (its terms are skipped)
#pragma unroll
    for (int r = 0; r < RT; r++) at[r][u] = src(r, jj[u]);
  }
#pragma unroll
  for (int u = 0; u < JU; u++) {
    const u64* cj = query(jj[u]);
    blk.c0[u] = *reinterpret_cast<const pir_u64x2*>(cj);
    blk.c1[u] = *reinterpret_cast<const pir_u64x2*>(cj + query.poly_stride);
#pragma unroll
    for (int r = 0; r < RT; r++) {
      const auto p = as_global(reinterpret_cast<const pir_u64x2*>(at[r][u]));  // global_load, not flat_load
      blk.pv[r][u] = __builtin_nontemporal_load(p);  // the database streams through once
    }
  }
}
template <int RT, int JU, class Query, class Src>
__device__ __forceinline__ void pir_dot_body(const DevMod& dm, u32 cols, const Query& query, const Src& src, u128 (&a0)[RT][2], u128 (&a1)[RT][2]) {
#pragma unroll
  for (int r = 0; r < RT; r++) a0[r][0] = a0[r][1] = a1[r][0] = a1[r][1] = 0;
  PirBlock<RT, JU> nxt;
  pir_fetch<RT, JU>(nxt, 0, cols, query, src);
  u32 since = 0;  // products accumulated since the last reduction: below 2^122 each, sixteen fit 128 bits
  for (u32 j0 = 0; j0 < cols; j0 += JU) {
    const PirBlock<RT, JU> cur = nxt;
    if (j0 + JU < cols) pir_fetch<RT, JU>(nxt, j0 + JU, cols, query, src);  // wave-uniform: in flight while `cur` is multiplied
#pragma unroll
    for (int u = 0; u < JU; u++) {
      const bool live = j0 + (u32)u < cols;  // wave-uniform; false only in a partial last block
      const u64 m = live ? ~0ull : 0ull;
#pragma unroll
      for (int r = 0; r < RT; r++) {
        const u64 px = cur.pv[r][u].x & m, py = cur.pv[r][u].y & m;
        a0[r][0] += (u128)cur.c0[u].x * px;
        a0[r][1] += (u128)cur.c0[u].y * py;
        a1[r][0] += (u128)cur.c1[u].x * px;
        a1[r][1] += (u128)cur.c1[u].y * py;
      }
    }
    since += JU;
    if (since + JU > 16u) {
      since = 0;
#pragma unroll
      for (int r = 0; r < RT; r++)
#pragma unroll
        for (int e = 0; e < 2; e++) a0[r][e] = reduce128_fast(a0[r][e], dm), a1[r][e] = reduce128_fast(a1[r][e], dm);
    }
  }
}
#define PIR_JU 1  // columns per block, two blocks in flight (measured, r04: JU = 1 / 2 / 4 -> the 16 GiB product 2.95 / 3.04 / 5.1 ms)
// Which grid dimension walks the ROW BLOCKS.  Workgroups are dispatched x-fastest, so with the row blocks on x (PIR_ROWS_FAST) the
// workgroups resident at one time are all row blocks of a few (coefficient range, residue) slices: they read the SAME query words
// at about the same time, and each XCD's L2 serves them after the first.  With the coefficient ranges on x (the r01-r03 order) the
// resident workgroups cover all coefficients of 8 row blocks, and every row block re-reads the whole transformed query: at
// n = 16384 that is 512 MiB per 4 rows -- 64 GiB beside the 128 GiB database, and more than the 256 MB Infinity Cache holds.
// r06 s31: ... and all row blocks of ONE slice on ONE XCD.  Workgroups go to the 8 XCDs round robin in dispatch order, so with the row
// blocks merely x-fastest a slice's row blocks were spread over all eight L2s: every XCD fetched the slice's query words from HBM, and
// with few row blocks (one GPU's share of a 1024 x 1024 database is 128 rows: 32 row blocks, four per XCD) the few that shared an L2
// drifted apart -- the 128 x 1024 product ran at 5.2 TB/s against 6.0 for 512 x 256.  Now dispatch index d = 8 q + xcd is mapped to
// row block q % RB of slice (q / RB) * 8 + xcd: an XCD walks its own slices, each slice's row blocks back to back on that XCD.
#ifndef PIR_XCD_SLICES
#define PIR_XCD_SLICES 1  // (experiment hook: 0 = the x-fastest order of r04 ... r06 s30)
#endif
struct PirGrid {
  u32 rb, xb, res;
  __device__ __forceinline__ PirGrid() {
    const u32 RB = gridDim.x, XB = gridDim.y, S = XB * gridDim.z;
    rb = blockIdx.x, xb = blockIdx.y, res = blockIdx.z;
    if (PIR_XCD_SLICES && (S & 7u) == 0u) {  // (wave-uniform; otherwise the plain x-fastest order)
      const u32 d = blockIdx.x + RB * (blockIdx.y + XB * blockIdx.z);
      const u32 q = d >> 3, sl = (q / RB) * 8u + (d & 7u);
      rb = q % RB, xb = sl % XB, res = sl / XB;
    }
  }
  __device__ __forceinline__ u32 rowblock() const { return rb; }
  __device__ __forceinline__ u32 xblock() const { return xb; }
  __device__ __forceinline__ u32 residue() const { return res; }
  static dim3 grid(u32 xblocks, u32 K, u32 rowblocks) { return dim3(rowblocks, xblocks, K); }
};

template <int RT>
__global__ __launch_bounds__(kClientThreads) void dot_plain2_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ ctn, u32 cols,
                                                                    const u64* __restrict__ pntt, u32 rows, u64* __restrict__ acc) {
  const u32 n = ctx->n, K = ctx->K;
  const PirGrid pg;
  const u32 x = 2 * (pg.xblock() * kClientThreads + threadIdx.x);
  const u32 i = pg.residue(), r0 = pg.rowblock() * RT;
  if (x >= n) return;
  const DevMod& dm = ctx->mod[i];
  const size_t in_row = (size_t)i * n + x;
  struct Query {
    const u64* base;
    size_t col_stride, poly_stride;
    __device__ __forceinline__ const u64* operator()(u32 j) const { return base + (size_t)j * col_stride; }
  } query{ctn + in_row, (size_t)2 * K * n, (size_t)K * n};
  struct Src {
    const u64* base;  // row r0, column 0
    size_t row_stride, col_stride;
    u32 last;         // rows - 1 - r0: rows of this thread's RT beyond it are clamped to the matrix's last row
    __device__ __forceinline__ const u64* operator()(int r, u32 j) const {
      return base + (size_t)((u32)r < last ? (u32)r : last) * row_stride + (size_t)j * col_stride;
    }
  } src{pntt + (size_t)r0 * cols * K * n + in_row, (size_t)cols * K * n, (size_t)K * n, rows - 1 - r0};
  u128 a0[RT][2], a1[RT][2];
  pir_dot_body<RT, PIR_JU>(dm, cols, query, src, a0, a1);
#pragma unroll
  for (int r = 0; r < RT; r++) {
    if (r0 + r < rows) {
      pir_u64x2 o0, o1;
      o0.x = reduce128_fast(a0[r][0], dm), o0.y = reduce128_fast(a0[r][1], dm);
      o1.x = reduce128_fast(a1[r][0], dm), o1.y = reduce128_fast(a1[r][1], dm);
      *reinterpret_cast<pir_u64x2*>(&acc[((((size_t)(r0 + r)) * 2 + 0) * K + i) * n + x]) = o0;
      *reinterpret_cast<pir_u64x2*>(&acc[((((size_t)(r0 + r)) * 2 + 1) * K + i) * n + x]) = o1;
    }
  }
}

// The graph executor's form (program.cpp: MultiplyPlaintext -> Add chains kept in the transform domain): the plaintexts of a
// (row, term) sit behind a descriptor table instead of one dense matrix -- program arguments are separate buffers -- and a batch
// dimension rides on grid z.  ctn u64[cols][batch][2][K][N]; tab [rows][cols]; acc u64[rows][batch][2][K][N].  The table reads are
// wave-uniform (scalar loads); everything else is dot_plain2_kernel.
template <int RT>
__global__ __launch_bounds__(kClientThreads) void dot_plain_tab_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ ctn, u32 cols,
                                                                       const PlainNttRef* __restrict__ tab, u32 rows, u32 batch, u64* __restrict__ acc) {
  const u32 n = ctx->n, K = ctx->K;
  const PirGrid pg;
  const u32 x = 2 * (pg.xblock() * kClientThreads + threadIdx.x);
  const u32 i = pg.residue(), r0 = (pg.rowblock() / batch) * RT, b = pg.rowblock() % batch;
  if (x >= n) return;
  const DevMod& dm = ctx->mod[i];
  const size_t in_row = (size_t)i * n + x;
  struct Query {
    const u64* base;
    size_t col_stride, poly_stride;
    __device__ __forceinline__ const u64* operator()(u32 j) const { return base + (size_t)j * col_stride; }
  } query{ctn + (size_t)b * 2 * K * n + in_row, (size_t)batch * 2 * K * n, (size_t)K * n};
  struct Src {
    const PlainNttRef* tab;  // row r0, column 0
    u32 cols, last, b;
    size_t in_row;
    __device__ __forceinline__ const u64* operator()(int r, u32 j) const {
      const PlainNttRef ref = tab[(size_t)((u32)r < last ? (u32)r : last) * cols + j];  // wave-uniform: a scalar load
      return ref.ptr + (size_t)b * ref.stride + in_row;
    }
  } src{tab + (size_t)r0 * cols, cols, rows - 1 - r0, b, in_row};
  u128 a0[RT][2], a1[RT][2];
  pir_dot_body<RT, PIR_JU>(dm, cols, query, src, a0, a1);
#pragma unroll
  for (int r = 0; r < RT; r++) {
    if (r0 + r < rows) {
      pir_u64x2 o0, o1;
      o0.x = reduce128_fast(a0[r][0], dm), o0.y = reduce128_fast(a0[r][1], dm);
      o1.x = reduce128_fast(a1[r][0], dm), o1.y = reduce128_fast(a1[r][1], dm);
      u64* dst = acc + ((size_t)(r0 + r) * batch + b) * 2 * K * n + in_row;
      *reinterpret_cast<pir_u64x2*>(dst) = o0;
      *reinterpret_cast<pir_u64x2*>(dst + (size_t)K * n) = o1;
    }
  }
}

// ---- launchers ----
hipError_t launch_dot_plain_tab(const DevCtx* ctx, u32 n, u32 K, const u64* ctn, u32 cols, const PlainNttRef* tab, u32 rows, u32 batch, u64* acc, hipStream_t s) {
  constexpr int RT = 4;
  // grid z <= 65535: row blocks go in slices
  if (batch == 0 || batch > 65535u) return hipErrorInvalidValue;  // one row block's items must fit grid z (callers chunk the batch)
  const u32 per = std::max(1u, 65535u / batch);  // row blocks per launch
  const u32 blocks = (rows + RT - 1) / RT;
  for (u32 off = 0; off < blocks; off += per) {
    const u32 c = std::min(per, blocks - off);
    const u32 r_off = off * RT, r_cnt = std::min(rows - r_off, c * RT);
    dot_plain_tab_kernel<RT><<<PirGrid::grid((n / 2 + kClientThreads - 1) / kClientThreads, K, c * batch), kClientThreads, 0, s>>>(ctx, ctn, cols, tab + (size_t)r_off * cols, r_cnt, batch,
                                                                                   acc + (size_t)r_off * batch * 2 * K * n);
  }
  return hipGetLastError();
}
hipError_t launch_keygen_ternary(const DevCtx* ctx, u32 n, const RngSeed& seed, u64 stream, u64* s_out, hipStream_t s) {
  keygen_ternary_kernel<<<cgrid(n, 1), kClientThreads, 0, s>>>(ctx, seed.secret, stream, s_out);
  return hipGetLastError();
}
hipError_t launch_keygen_sample(const DevCtx* ctx, u32 n, const RngSeed& seed, u64 stream0, u64* a, u64* e, size_t count, hipStream_t s) {
  keygen_sample_kernel<<<cgrid(n, (u32)count), kClientThreads, 0, s>>>(ctx, seed, stream0, a, e);
  return hipGetLastError();
}
hipError_t launch_keygen_assemble(const DevCtx* ctx, u32 n, u32 KK, const u64* a, const u64* e, const u64* sk, const u64* w, u64* key, size_t count,
                                  hipStream_t s) {
  keygen_assemble_kernel<<<cgrid(n, KK, (u32)count), kClientThreads, 0, s>>>(ctx, a, e, sk, w, key);
  return hipGetLastError();
}
hipError_t launch_keygen_square(const DevCtx* ctx, u32 n, u32 KK, const u64* in, u64* out, hipStream_t s) {
  keygen_square_kernel<<<cgrid(n, KK), kClientThreads, 0, s>>>(ctx, in, out);
  return hipGetLastError();
}
hipError_t launch_keygen_galois(const DevCtx* ctx, u32 n, u32 KK, const u64* in, u64* out, u32 ginv, hipStream_t s) {
  keygen_galois_kernel<<<cgrid(n, KK), kClientThreads, 0, s>>>(ctx, in, out, ginv);
  return hipGetLastError();
}
hipError_t launch_crt_compose(const DevCtx* ctx, u32 n, u32 KC, const u64* consts, const u64* in, u64* out, u32 polys, hipStream_t s) {
  if (!polys) return hipSuccess;
#define HB_CRT(K_)                                                                                 \
  case K_:                                                                                         \
    crt_compose_kernel<K_><<<cgrid(n, polys), kClientThreads, 0, s>>>(ctx, consts, in, out);       \
    break;
  switch (KC) {
    HB_CRT(1) HB_CRT(2) HB_CRT(3) HB_CRT(4) HB_CRT(5) HB_CRT(6) HB_CRT(7) HB_CRT(8)
    HB_CRT(9) HB_CRT(10) HB_CRT(11) HB_CRT(12) HB_CRT(13) HB_CRT(14) HB_CRT(15) HB_CRT(16)
    default:
      return hipErrorInvalidValue;
  }
#undef HB_CRT
  return hipGetLastError();
}
hipError_t launch_crt_decompose(const DevCtx* ctx, u32 n, u32 KC, const u64* in, u64* out, u32 polys, hipStream_t s) {
  if (!polys) return hipSuccess;
  crt_decompose_kernel<<<cgrid(n, KC, polys), kClientThreads, 0, s>>>(ctx, KC, in, out);
  return hipGetLastError();
}
// 1 (default): two coefficients and 4 rows per thread -- 16-byte accesses; measured (interleaved A/B, 256 x 256 entries, n = 8192):
// the product kernel 4.89 -> 3.42 ms, i.e. the 16 GiB database streams at 5.0 TB/s instead of 3.5, 10.5 M -> 13.7 M entries/s.
// 2: two coefficients and 8 rows (128 registers of accumulators): 4.41 ms.  0: one coefficient, 8 rows, 8-byte accesses.
hipError_t launch_dot_plain(const DevCtx* ctx, u32 n, u32 K, const u64* ctn, u32 cols, const u64* pntt, u32 rows, u64* acc, hipStream_t s) {
  constexpr int RT = 4;
  dot_plain2_kernel<RT><<<PirGrid::grid((n / 2 + kClientThreads - 1) / kClientThreads, K, (rows + RT - 1) / RT), kClientThreads, 0, s>>>(ctx, ctn, cols, pntt, rows, acc);
  return hipGetLastError();
}
hipError_t launch_batch_scatter(const DevCtx* ctx, u32 n, const u32* map, const u64* values, u64* plain, size_t ops, int is_signed, u32* bad,
                                hipStream_t s) {
  batch_scatter_kernel<<<cgrid(n, (u32)ops), kClientThreads, 0, s>>>(ctx, map, values, plain, is_signed, bad);
  return hipGetLastError();
}
hipError_t launch_batch_gather(const DevCtx* ctx, u32 n, const u32* map, const u64* tmp, u64* values, size_t ops, int is_signed, hipStream_t s) {
  batch_gather_kernel<<<cgrid(n, (u32)ops), kClientThreads, 0, s>>>(ctx, map, tmp, values, is_signed);
  return hipGetLastError();
}
hipError_t launch_dot_secret(const DevCtx* ctx, u32 n, u32 K, const u64* ctn, u32 size, const u64* sk, u64* acc, size_t ops, hipStream_t s) {
  dot_secret_kernel<<<cgrid(n, K, (u32)ops), kClientThreads, 0, s>>>(ctx, ctn, size, sk, acc);
  return hipGetLastError();
}
hipError_t launch_decrypt_round(const DevCtx* ctx, u32 n, const u64* ct, u32 size, const u64* acc, u64* plain, size_t ops, hipStream_t s) {
  decrypt_round_kernel<<<cgrid(n, (u32)ops), kClientThreads, 0, s>>>(ctx, ct, size, acc, plain);
  return hipGetLastError();
}
hipError_t launch_encrypt_sample(const DevCtx* ctx, u32 n, const RngSeed& seed, u64 op0, u64* u, u64* e, size_t ops, hipStream_t s) {
  encrypt_sample_kernel<<<cgrid(n, (u32)ops), kClientThreads, 0, s>>>(ctx, seed.secret, op0, u, e);
  return hipGetLastError();
}
hipError_t launch_encrypt_finish(const DevCtx* ctx, u32 n, const RngSeed& seed, u64 op0, const u64* c2, const u64* plain, size_t pstride, u64* out, size_t ops,
                                 hipStream_t s) {
  encrypt_finish_kernel<<<cgrid(n, (u32)ops), kClientThreads, 0, s>>>(ctx, seed.secret, op0, c2, plain, pstride, out);
  return hipGetLastError();
}
hipError_t launch_encrypt_dyadic(const DevCtx* ctx, u32 n, u32 KK, const u64* un, const u64* pk, u64* c, size_t ops, hipStream_t s) {
  encrypt_dyadic_kernel<<<cgrid(n, KK, (u32)ops), kClientThreads, 0, s>>>(ctx, un, pk, c);
  return hipGetLastError();
}
hipError_t launch_add_key_level(const DevCtx* ctx, u32 n, u64* c, const u64* e, size_t residue_polys, hipStream_t s) {
  add_key_level_kernel<<<cgrid(n, (u32)residue_polys), kClientThreads, 0, s>>>(ctx, c, e);
  return hipGetLastError();
}

}  // namespace hipbfv
