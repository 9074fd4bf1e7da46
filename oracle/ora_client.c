/*
 * oracle/ora_client.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Minimal key generation, encryption, decryption and batch encoding so that the parity tests can
 * run the reference's decrypt-and-compare test matrix (seal_fhe/src/bfv_evaluator.rs:255-970,
 * seal_fhe/tests/assumptions.rs) against the oracle and against the HIP path.  These steps sit on
 * either side of the hot path and are NOT part of the accelerated product.
 * Restates SEAL 4.0 keygenerator.cpp / encryptor.cpp / decryptor.cpp / batchencoder.cpp /
 * util/rlwe.cpp; the random sampling is statistically, not bit-wise, equivalent (SEAL's own output
 * is random; nothing in the reference pins it outside the `deterministic` feature hash).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "ora_internal.h"

/* ------------------------------------------------------------------ PRNG (xoshiro256**) */

static uint64_t g_s[4] = {0x9E3779B97F4A7C15ull, 0xBF58476D1CE4E5B9ull, 0x94D049BB133111EBull, 0x2545F4914F6CDD1Dull};

static uint64_t splitmix(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void ora_seed(uint64_t seed)
{
    for (int i = 0; i < 4; i++) g_s[i] = splitmix(&seed);
}

static inline uint64_t rotl(uint64_t x, int k)
{
    return (x << k) | (x >> (64 - k));
}

uint64_t ora_rand64(void)
{
    uint64_t r = rotl(g_s[1] * 5, 7) * 9, t = g_s[1] << 17;
    g_s[2] ^= g_s[0];
    g_s[3] ^= g_s[1];
    g_s[1] ^= g_s[2];
    g_s[0] ^= g_s[3];
    g_s[2] ^= t;
    g_s[3] = rotl(g_s[3], 45);
    return r;
}

static uint64_t rand_below(uint64_t q)
{
    uint64_t mask = ~(uint64_t)0 >> __builtin_clzll(q);
    for (;;) {
        uint64_t v = ora_rand64() & mask;
        if (v < q) return v;
    }
}

/* small signed samples, stored as int8 */
static void sample_ternary(int8_t *s, size_t n)
{
    for (size_t k = 0; k < n; k++) s[k] = (int8_t)((int)rand_below(3) - 1);
}

/* clipped rounded Gaussian, sigma 3.2, |x| <= 19 (SEAL_USE_GAUSSIAN_NOISE=ON: seal_fhe/build.rs:50) */
static void sample_noise(int8_t *e, size_t n)
{
    for (size_t k = 0; k < n; k++) {
        for (;;) {
            double u1 = ((double)(ora_rand64() >> 11) + 1.0) / 9007199254740993.0;
            double u2 = (double)(ora_rand64() >> 11) / 9007199254740992.0;
            double z = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2) * 3.2;
            if (fabs(z) <= 19.2) {
                e[k] = (int8_t)lround(z);
                break;
            }
        }
    }
}

static void small_to_rns(const ora_ctx *c, const int8_t *s, size_t nres, uint64_t *out)
{
    const size_t n = c->n;
    for (size_t i = 0; i < nres; i++) {
        const uint64_t q = c->key_mod[i].q;
        for (size_t k = 0; k < n; k++) out[i * n + k] = s[k] < 0 ? q - (uint64_t)(-s[k]) : (uint64_t)s[k];
    }
}

/* ------------------------------------------------------------------ key generation */

void ora_keygen_secret(const ora_ctx *c, uint64_t *sk_ntt)
{
    const size_t n = c->n, KK = c->key_count;
    int8_t *s = (int8_t *)malloc(n);
    sample_ternary(s, n);
    small_to_rns(c, s, KK, sk_ntt);
    for (size_t i = 0; i < KK; i++) ora_ntt_fwd(&c->key_ntt[i], sk_ntt + i * n);
    free(s);
}

/* (c0, c1) = (-(a*s + e), a) over the key level, NTT form: out [2][KK][n] */
static void encrypt_zero_symmetric_keylevel_ntt(const ora_ctx *c, const uint64_t *sk_ntt, uint64_t *out)
{
    const size_t n = c->n, KK = c->key_count;
    int8_t *e = (int8_t *)malloc(n);
    sample_noise(e, n);
    uint64_t *en = (uint64_t *)malloc(KK * n * sizeof(uint64_t));
    small_to_rns(c, e, KK, en);
    for (size_t i = 0; i < KK; i++) {
        const ora_mod *m = &c->key_mod[i];
        ora_ntt_fwd(&c->key_ntt[i], en + i * n);
        uint64_t *c0 = out + i * n, *c1 = out + (KK + i) * n;
        for (size_t k = 0; k < n; k++) {
            c1[k] = rand_below(m->q);
            uint64_t v = ora_addmod(ora_mulmod(c1[k], sk_ntt[i * n + k], m), en[i * n + k], m);
            c0[k] = ora_negmod(v, m);
        }
    }
    free(en);
    free(e);
}

void ora_keygen_public(const ora_ctx *c, const uint64_t *sk_ntt, uint64_t *pk)
{
    encrypt_zero_symmetric_keylevel_ntt(c, sk_ntt, pk);
}

/* SEAL KeyGenerator::generate_one_kswitch_key: key[J] = Enc_sym(0) with q_sp * newkey added to
 * residue J of component 0.  Flat layout u64[K][2][K+1][N] (seal_fhe/src/key_generator.rs:467-631). */
void ora_keygen_kswitch(const ora_ctx *c, const uint64_t *sk_ntt, const uint64_t *newkey_ntt, uint64_t *key)
{
    const size_t n = c->n, KK = c->key_count, K = c->K;
    const uint64_t qsp = c->key_mod[KK - 1].q;
    for (size_t J = 0; J < K; J++) {
        uint64_t *kj = key + J * 2 * KK * n;
        encrypt_zero_symmetric_keylevel_ntt(c, sk_ntt, kj);
        const ora_mod *m = &c->key_mod[J];
        uint64_t factor = ora_reduce64(qsp, m);
        uint64_t *dst = kj + J * n; /* component 0, residue J */
        for (size_t k = 0; k < n; k++) dst[k] = ora_addmod(dst[k], ora_mulmod(newkey_ntt[J * n + k], factor, m), m);
    }
}

void ora_keygen_relin(const ora_ctx *c, const uint64_t *sk_ntt, uint64_t *rk)
{
    const size_t n = c->n, KK = c->key_count;
    uint64_t *s2 = (uint64_t *)malloc(KK * n * sizeof(uint64_t));
    for (size_t i = 0; i < KK; i++)
        for (size_t k = 0; k < n; k++) s2[i * n + k] = ora_mulmod(sk_ntt[i * n + k], sk_ntt[i * n + k], &c->key_mod[i]);
    ora_keygen_kswitch(c, sk_ntt, s2, rk);
    free(s2);
}

void ora_keygen_galois(const ora_ctx *c, const uint64_t *sk_ntt, uint32_t elt, uint64_t *key)
{
    const uint32_t n = c->n;
    const size_t KK = c->key_count;
    uint64_t *rot = (uint64_t *)malloc(KK * n * sizeof(uint64_t));
    uint64_t *tmp = (uint64_t *)malloc(n * sizeof(uint64_t));
    for (size_t i = 0; i < KK; i++) {
        const uint64_t q = c->key_mod[i].q;
        memcpy(tmp, sk_ntt + i * n, n * sizeof(uint64_t));
        ora_ntt_inv(&c->key_ntt[i], tmp);
        uint64_t raw = 0;
        for (uint32_t k = 0; k < n; k++, raw += elt) {
            uint32_t idx = (uint32_t)(raw & (n - 1));
            uint64_t v = tmp[k];
            if ((raw >> c->logn) & 1) v = v ? q - v : 0;
            rot[i * n + idx] = v;
        }
        ora_ntt_fwd(&c->key_ntt[i], rot + i * n);
    }
    ora_keygen_kswitch(c, sk_ntt, rot, key);
    free(tmp);
    free(rot);
}

/* SEAL GaloisTool::get_elts_all (seal_fhe/src/key_generator.rs:170-182 -> CreateGaloisKeysAll) */
size_t ora_galois_elts_all(const ora_ctx *c, uint32_t *out)
{
    const uint32_t m = 2 * c->n;
    size_t cnt = 0;
    out[cnt++] = m - 1;
    uint64_t pos = 3, neg = 0;
    /* inverse of 3 mod m (m a power of two) */
    {
        uint64_t x = 1;
        for (int i = 0; i < 6; i++) x = x * (2 - 3 * x);
        neg = x & (m - 1);
    }
    for (int i = 0; i < c->logn - 1; i++) {
        out[cnt++] = (uint32_t)pos;
        pos = (pos * pos) & (m - 1);
        out[cnt++] = (uint32_t)neg;
        neg = (neg * neg) & (m - 1);
    }
    return cnt;
}

/* ------------------------------------------------------------------ encryption */

/* SEAL RNSTool::divide_and_round_q_last_inplace: key level [KK][n] -> data level [K][n] */
static void divide_round_q_last(const ora_ctx *c, uint64_t *x, uint64_t *out)
{
    const size_t n = c->n, KK = c->key_count, K = c->K;
    const ora_mod *ml = &c->key_mod[KK - 1];
    const uint64_t half = ml->q >> 1;
    uint64_t *last = x + (KK - 1) * n;
    for (size_t k = 0; k < n; k++) last[k] = ora_addmod(last[k], half, ml);
    for (size_t i = 0; i < K; i++) {
        const ora_mod *m = &c->key_mod[i];
        const uint64_t half_mod = ora_reduce64(half, m);
        for (size_t k = 0; k < n; k++) {
            uint64_t tk = ora_submod(ora_reduce64(last[k], m), half_mod, m);
            uint64_t d = ora_submod(x[i * n + k], tk, m);
            out[i * n + k] = ora_mulop_mod(d, &c->inv_q_last_mod_q[i], m->q);
        }
    }
}

static void add_plain_scaled(const ora_ctx *c, const uint64_t *plain, size_t pc, uint64_t *ct2)
{
    ora_add_plain(c, ct2, 2, plain, pc, ct2);
}

void ora_encrypt(const ora_ctx *c, const uint64_t *pk, const uint64_t *plain, size_t pc, uint64_t *ct2)
{
    const size_t n = c->n, KK = c->key_count, K = c->K;
    int8_t *u = (int8_t *)malloc(n), *e = (int8_t *)malloc(n);
    uint64_t *un = (uint64_t *)malloc(KK * n * sizeof(uint64_t));
    uint64_t *cj = (uint64_t *)malloc(KK * n * sizeof(uint64_t));
    uint64_t *en = (uint64_t *)malloc(KK * n * sizeof(uint64_t));
    sample_ternary(u, n);
    small_to_rns(c, u, KK, un);
    for (size_t i = 0; i < KK; i++) ora_ntt_fwd(&c->key_ntt[i], un + i * n);
    for (size_t j = 0; j < 2; j++) {
        sample_noise(e, n);
        small_to_rns(c, e, KK, en);
        for (size_t i = 0; i < KK; i++) {
            const ora_mod *m = &c->key_mod[i];
            for (size_t k = 0; k < n; k++) cj[i * n + k] = ora_mulmod(un[i * n + k], pk[(j * KK + i) * n + k], m);
            ora_ntt_inv(&c->key_ntt[i], cj + i * n);
            for (size_t k = 0; k < n; k++) cj[i * n + k] = ora_addmod(cj[i * n + k], en[i * n + k], m);
        }
        if (KK > 1)
            divide_round_q_last(c, cj, ct2 + j * K * n);
        else
            memcpy(ct2 + j * K * n, cj, n * sizeof(uint64_t));
    }
    add_plain_scaled(c, plain, pc, ct2);
    free(u);
    free(e);
    free(un);
    free(cj);
    free(en);
}

void ora_encrypt_symmetric(const ora_ctx *c, const uint64_t *sk_ntt, const uint64_t *plain, size_t pc, uint64_t *ct2)
{
    const size_t n = c->n, KK = c->key_count, K = c->K;
    uint64_t *z = (uint64_t *)malloc(2 * KK * n * sizeof(uint64_t));
    encrypt_zero_symmetric_keylevel_ntt(c, sk_ntt, z);
    for (size_t j = 0; j < 2; j++) {
        for (size_t i = 0; i < KK; i++) ora_ntt_inv(&c->key_ntt[i], z + (j * KK + i) * n);
        if (KK > 1)
            divide_round_q_last(c, z + j * KK * n, ct2 + j * K * n);
        else
            memcpy(ct2 + j * K * n, z + j * KK * n, n * sizeof(uint64_t));
    }
    add_plain_scaled(c, plain, pc, ct2);
    free(z);
}

/* ------------------------------------------------------------------ decryption */

void ora_dot_with_secret(const ora_ctx *c, const uint64_t *ct, size_t s, const uint64_t *sk_ntt, uint64_t *out)
{
    const size_t n = c->n, K = c->K;
    uint64_t *tmp = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint64_t *spow = (uint64_t *)malloc(n * sizeof(uint64_t));
    for (size_t i = 0; i < K; i++) {
        const ora_mod *m = &c->key_mod[i];
        uint64_t *acc = out + i * n;
        memset(acc, 0, n * sizeof(uint64_t));
        memcpy(spow, sk_ntt + i * n, n * sizeof(uint64_t));
        for (size_t p = 1; p < s; p++) {
            memcpy(tmp, ct + (p * K + i) * n, n * sizeof(uint64_t));
            ora_ntt_fwd(&c->key_ntt[i], tmp);
            for (size_t k = 0; k < n; k++) acc[k] = ora_addmod(acc[k], ora_mulmod(tmp[k], spow[k], m), m);
            for (size_t k = 0; k < n; k++) spow[k] = ora_mulmod(spow[k], sk_ntt[i * n + k], m);
        }
        ora_ntt_inv(&c->key_ntt[i], acc);
        for (size_t k = 0; k < n; k++) acc[k] = ora_addmod(acc[k], ct[i * n + k], m);
    }
    free(tmp);
    free(spow);
}

/* SEAL RNSTool::decrypt_scale_and_round */
void ora_decrypt(const ora_ctx *c, const uint64_t *ct, size_t s, const uint64_t *sk_ntt, uint64_t *plain_out)
{
    const size_t n = c->n, K = c->K;
    uint64_t *d = (uint64_t *)malloc(K * n * sizeof(uint64_t));
    uint64_t *tg = (uint64_t *)malloc(2 * n * sizeof(uint64_t));
    ora_dot_with_secret(c, ct, s, sk_ntt, d);
    for (size_t i = 0; i < K; i++)
        for (size_t k = 0; k < n; k++) d[i * n + k] = ora_mulop_mod(d[i * n + k], &c->prod_t_gamma_mod_q[i], c->key_mod[i].q);
    ora_baseconv_apply(&c->q_to_tgamma, d, tg, n);
    const uint64_t t = c->t.q, gamma = c->gamma.q, gamma_half = gamma >> 1;
    for (size_t k = 0; k < n; k++) {
        uint64_t a = ora_mulop_mod(tg[k], &c->neg_inv_q_mod_t_gamma[0], t);
        uint64_t g = ora_mulop_mod(tg[n + k], &c->neg_inv_q_mod_t_gamma[1], gamma);
        uint64_t r;
        if (g > gamma_half)
            r = ora_addmod(a, ora_reduce64(gamma - g, &c->t), &c->t);
        else
            r = ora_submod(a, ora_reduce64(g, &c->t), &c->t);
        if (r) r = ora_mulop_mod(r, &c->inv_gamma_mod_t, t);
        plain_out[k] = r;
    }
    free(d);
    free(tg);
}

/* ------------------------------------------------------------------ BatchEncoder */

int ora_batch_encode(const ora_ctx *c, const uint64_t *values, uint64_t *plain)
{
    if (!c->t_batching) return -1;
    const size_t n = c->n;
    for (size_t i = 0; i < n; i++) {
        if (values[i] >= c->t.q) return -1;
        plain[c->batch_index_map[i]] = values[i];
    }
    ora_ntt_inv(&c->t_ntt, plain);
    return 0;
}

int ora_batch_decode(const ora_ctx *c, const uint64_t *plain, uint64_t *values)
{
    if (!c->t_batching) return -1;
    const size_t n = c->n;
    uint64_t *tmp = (uint64_t *)malloc(n * sizeof(uint64_t));
    memcpy(tmp, plain, n * sizeof(uint64_t));
    ora_ntt_fwd(&c->t_ntt, tmp);
    for (size_t i = 0; i < n; i++) values[i] = tmp[c->batch_index_map[i]];
    free(tmp);
    return 0;
}
