/*
 * oracle/bfv_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("oracle") of the BFV ciphertext-arithmetic hot path that the reference
 * executes inside Microsoft SEAL 4.0 behind seal_fhe::Evaluator
 * (seal_fhe/src/evaluator.rs:7-280, evaluator_base.rs:55-407, bfv_evaluator.rs:12-248).
 *
 * PARITY STATUS: SEAL's source is not present in /root/reference (seal_fhe/SEAL/ is an empty
 * submodule; pinned only as "lib version 4.0", seal_fhe/build.rs:84-85), so this oracle
 * restates SEAL 4.0's published algorithms (Harvey NTT, BEHZ RNS multiply, hybrid
 * key-switching with one special prime) and is PINNED against:
 *   - the reference's prime-generation known answers (seal_fhe/src/modulus.rs:279-313,
 *     encryption_parameters.rs:340-365, logproof/src/rings.rs:36-125),
 *   - the reference's binary SEAL key fixtures (seal_fhe/tests/data/{secret,public}_key.bin),
 *     which pin the NTT root choice, ordering and [poly][rns][coeff] layout bit-for-bit,
 *   - every decrypt-and-compare evaluator test (seal_fhe/src/bfv_evaluator.rs:322-970,
 *     seal_fhe/tests/assumptions.rs, sunscreen_runtime/src/run.rs:546-882).
 * Ciphertext-bit parity of multiply/relinearize/rotate with real SEAL is NOT pinned by any
 * reference test ("parity unpinned" for those bits).  What stands in for it
 * on multiply: tests/test_oracle_behz_exact.py carries BEHZ out over the integers (no auxiliary
 * base) and gets this library's bits, random and extreme operands, n = 1024 ... 16384;
 * tests/test_oracle_keyswitch_exact.py / test_oracle_plain_exact.py do the same for relinearize, apply_galois,
 * mod_switch_to_next and the plaintext operations.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 *
 * Layouts (seal_fhe/src/plaintext_ciphertext.rs:303-314): ciphertext = u64[size][K][N];
 * key-switching key = u64[K][2][K+1][N] in NTT form, special prime last.
 */
#ifndef BFV_ORACLE_H
#define BFV_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_ctx ora_ctx;

/* ---- number theory (SEAL util/numth, CoeffModulus, PlainModulus) ---- */
int ora_is_prime(uint64_t v);
/* primes p == 1 (mod factor), descending from the largest such value < 2^bits; returns count found */
size_t ora_get_primes(uint64_t factor, int bits, size_t count, uint64_t *out);
/* CoeffModulus::Create(n, bit_sizes): seal_fhe/src/modulus.rs:149-180 */
int ora_coeff_modulus_create(uint32_t n, const int *bit_sizes, size_t count, uint64_t *out);
/* PlainModulus::Batching(n, bits): seal_fhe/src/modulus.rs:100-116 */
uint64_t ora_plain_batching(uint32_t n, int bits);
/* CoeffModulus::BFVDefault(n, sec in {128,192,256}); returns count (0 if unsupported) */
size_t ora_bfv_default(uint32_t n, int sec, uint64_t *out);
/* minimal primitive 2n-th root of unity mod q (0 if none) */
uint64_t ora_minimal_primitive_root(uint32_t two_n, uint64_t q);

/* ---- context ---- */
/* coeff_modulus: key-level primes (last one is the special prime when count > 1). */
ora_ctx *ora_ctx_create(uint32_t n, const uint64_t *coeff_modulus, size_t count, uint64_t plain_modulus);
void ora_ctx_destroy(ora_ctx *c);
uint32_t ora_ctx_n(const ora_ctx *c);
size_t ora_ctx_K(const ora_ctx *c);          /* data-level prime count        */
size_t ora_ctx_key_count(const ora_ctx *c);  /* key-level prime count         */
uint64_t ora_ctx_prime(const ora_ctx *c, size_t i); /* key-level prime i       */
uint64_t ora_ctx_plain(const ora_ctx *c);
size_t ora_ctx_bsk_count(const ora_ctx *c);
uint64_t ora_ctx_bsk_prime(const ora_ctx *c, size_t j); /* B primes then m_sk   */
uint64_t ora_ctx_gamma(const ora_ctx *c);
int ora_ctx_total_coeff_bits(const ora_ctx *c);

/* ---- NTT over key-level prime `idx` (in place, canonical in / canonical out) ---- */
void ora_ntt_forward(const ora_ctx *c, size_t idx, uint64_t *x);
void ora_ntt_inverse(const ora_ctx *c, size_t idx, uint64_t *x);
/* NTT over Bsk prime j */
void ora_ntt_forward_bsk(const ora_ctx *c, size_t j, uint64_t *x);
void ora_ntt_inverse_bsk(const ora_ctx *c, size_t j, uint64_t *x);
/* NTT over the plain modulus (batching); returns -1 if t != 1 mod 2n */
int ora_ntt_forward_plain(const ora_ctx *c, uint64_t *x);
int ora_ntt_inverse_plain(const ora_ctx *c, uint64_t *x);

/* ---- evaluator (a1..a5 of SURVEY section 8a). All return 0 on success, <0 on error. ---- */
#define ORA_E_INVALIDARG (-1)
#define ORA_E_TRANSPARENT (-2)
#define ORA_E_NOKEY (-3)

int ora_mod_switch_to_next(const ora_ctx *c, const uint64_t *ct, size_t s, uint64_t *out);
int ora_add(const ora_ctx *c, const uint64_t *a, size_t sa, const uint64_t *b, size_t sb, uint64_t *out);
int ora_sub(const ora_ctx *c, const uint64_t *a, size_t sa, const uint64_t *b, size_t sb, uint64_t *out);
int ora_negate(const ora_ctx *c, const uint64_t *a, size_t sa, uint64_t *out);
/* out has sa+sb-1 polys */
int ora_multiply(const ora_ctx *c, const uint64_t *a, size_t sa, const uint64_t *b, size_t sb, uint64_t *out);
/* ct: size 3 in, out: size 2. rk = u64[K][2][K+1][N] */
int ora_relinearize(const ora_ctx *c, const uint64_t *ct3, const uint64_t *rk, uint64_t *out2);
/* ct (size 2, in/out) += switch_key(target, key) */
int ora_switch_key(const ora_ctx *c, uint64_t *ct2, const uint64_t *target, const uint64_t *key);
/* coefficient-domain automorphism of one RNS poly (K residues) */
int ora_apply_galois_poly(const ora_ctx *c, const uint64_t *in, uint32_t galois_elt, uint64_t *out);
uint32_t ora_galois_elt_from_step(const ora_ctx *c, int step);
/* gk: array of 'n' pointers indexed by (elt-1)/2 (NULL = key missing) */
int ora_apply_galois(const ora_ctx *c, const uint64_t *ct2, uint32_t galois_elt,
                     const uint64_t *const *gk, uint64_t *out2);
int ora_rotate_rows(const ora_ctx *c, const uint64_t *ct2, int steps, const uint64_t *const *gk, uint64_t *out2);
int ora_rotate_columns(const ora_ctx *c, const uint64_t *ct2, const uint64_t *const *gk, uint64_t *out2);
/* plain: coeffs mod t, plain_count <= N */
int ora_add_plain(const ora_ctx *c, const uint64_t *ct, size_t s, const uint64_t *plain, size_t plain_count, uint64_t *out);
int ora_sub_plain(const ora_ctx *c, const uint64_t *ct, size_t s, const uint64_t *plain, size_t plain_count, uint64_t *out);
int ora_multiply_plain(const ora_ctx *c, const uint64_t *ct, size_t s, const uint64_t *plain, size_t plain_count, uint64_t *out);

/* ---- BEHZ building blocks exposed for unit tests ---- */
/* in: u64[K][N] (base q) -> out: u64[Bsk][N] after fastbconv_m_tilde + sm_mrq */
void ora_behz_extend(const ora_ctx *c, const uint64_t *in_q, uint64_t *out_bsk);
/* in: u64[K+Bsk][N] -> out u64[K][N]: fast_floor then fastbconv_sk */
void ora_behz_floor_sk(const ora_ctx *c, const uint64_t *in_q_bsk, uint64_t *out_q);

/* ---- client side (keygen/encrypt/decrypt/encode): needed only so tests can decrypt ---- */
void ora_seed(uint64_t seed);
void ora_keygen_secret(const ora_ctx *c, uint64_t *sk_ntt /* [K+1][N] */);
void ora_keygen_public(const ora_ctx *c, const uint64_t *sk_ntt, uint64_t *pk /* [2][K+1][N] */);
/* switching key for new key polynomial newkey_ntt ([K+1][N], NTT form) */
void ora_keygen_kswitch(const ora_ctx *c, const uint64_t *sk_ntt, const uint64_t *newkey_ntt, uint64_t *key);
void ora_keygen_relin(const ora_ctx *c, const uint64_t *sk_ntt, uint64_t *rk);
void ora_keygen_galois(const ora_ctx *c, const uint64_t *sk_ntt, uint32_t galois_elt, uint64_t *key);
/* all galois elements SEAL's create_galois_keys() makes; returns count */
size_t ora_galois_elts_all(const ora_ctx *c, uint32_t *out);
void ora_encrypt(const ora_ctx *c, const uint64_t *pk, const uint64_t *plain, size_t plain_count, uint64_t *ct2);
void ora_encrypt_symmetric(const ora_ctx *c, const uint64_t *sk_ntt, const uint64_t *plain, size_t plain_count, uint64_t *ct2);
/* c0 + c1 s + ... mod q_i, coefficient form: out u64[K][N] */
void ora_dot_with_secret(const ora_ctx *c, const uint64_t *ct, size_t s, const uint64_t *sk_ntt, uint64_t *out);
/* BFV decrypt (SEAL decrypt_scale_and_round via the {t,gamma} base): out u64[N] */
void ora_decrypt(const ora_ctx *c, const uint64_t *ct, size_t s, const uint64_t *sk_ntt, uint64_t *plain_out);
/* BatchEncoder: values[N] (mod t) <-> plain[N] */
int ora_batch_encode(const ora_ctx *c, const uint64_t *values, uint64_t *plain);
int ora_batch_decode(const ora_ctx *c, const uint64_t *plain, uint64_t *values);

/* ---- timing helpers for bench.py's cpu_baseline leg ---- */
/* runs `count` mul+relin ops over inputs a,b (each u64[count][2][K][N]) writing out u64[count][2][K][N];
   uses `threads` OpenMP threads if built with -fopenmp (else 1). returns seconds. */
double ora_bench_mul_relin(const ora_ctx *c, const uint64_t *a, const uint64_t *b, const uint64_t *rk,
                           uint64_t *out, size_t count, int threads);
/* forward+inverse NTT of `count` residue polys over key prime (i mod nprimes) ; returns seconds */
double ora_bench_ntt(const ora_ctx *c, uint64_t *x, size_t count, size_t nprimes, int threads);

#ifdef __cplusplus
}
#endif
#endif
