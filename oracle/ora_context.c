/*
 * oracle/ora_context.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Number theory, NTT tables and the BFV context/RNS-tool constants, restating SEAL 4.0
 * (util/numth.cpp, util/ntt.cpp, util/rns.cpp, util/globals.cpp, context.cpp; source absent
 * from /root/reference -- see bfv_oracle.h).  Reference call sites that build these objects:
 * seal_fhe/src/context.rs:63-80 (SEALContext_Create), seal_fhe/src/modulus.rs:100-180
 * (PlainModulus::batching, CoeffModulus::create / bfv_default).
 */
#include <stdlib.h>
#include <string.h>
#include "ora_internal.h"

/* ------------------------------------------------------------------ numth */

int ora_is_prime(uint64_t v)
{
    if (v < 2) return 0;
    static const uint64_t small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (size_t i = 0; i < sizeof(small) / sizeof(small[0]); i++) {
        if (v == small[i]) return 1;
        if (v % small[i] == 0) return 0;
    }
    ora_mod m;
    ora_mod_init(&m, v);
    uint64_t d = v - 1;
    int r = 0;
    while (!(d & 1)) {
        d >>= 1;
        r++;
    }
    /* deterministic Miller-Rabin for 64-bit integers */
    for (size_t i = 0; i < sizeof(small) / sizeof(small[0]); i++) {
        uint64_t x = ora_powmod(small[i], d, &m);
        if (x == 1 || x == v - 1) continue;
        int comp = 1;
        for (int k = 1; k < r; k++) {
            x = ora_mulmod(x, x, &m);
            if (x == v - 1) {
                comp = 0;
                break;
            }
        }
        if (comp) return 0;
    }
    return 1;
}

/* SEAL util::get_primes(factor, bit_size, count): start at the largest value == 1 (mod factor)
 * below 2^bit_size and walk down by `factor`. Pinned by seal_fhe/src/modulus.rs:279-313. */
size_t ora_get_primes(uint64_t factor, int bits, size_t count, uint64_t *out)
{
    if (bits < 2 || bits > 62 || factor == 0) return 0;
    uint64_t value = ((((uint64_t)1) << bits) - 1) / factor * factor + 1;
    uint64_t lower = ((uint64_t)1) << (bits - 1);
    size_t found = 0;
    while (found < count && value > lower) {
        if (ora_is_prime(value)) out[found++] = value;
        if (value < factor) break;
        value -= factor;
    }
    return found;
}

/* CoeffModulus::Create: per distinct bit size take the needed number of primes from the
 * descending list and hand them out smallest-first (seal_fhe/src/encryption_parameters.rs:340-365). */
int ora_coeff_modulus_create(uint32_t n, const int *bit_sizes, size_t count, uint64_t *out)
{
    if (count == 0 || count > ORA_MAXP) return -1;
    int done[ORA_MAXP] = {0};
    for (size_t i = 0; i < count; i++) {
        if (done[i]) continue;
        size_t need = 0;
        for (size_t j = i; j < count; j++)
            if (bit_sizes[j] == bit_sizes[i]) need++;
        uint64_t tmp[ORA_MAXP];
        if (ora_get_primes(2ull * n, bit_sizes[i], need, tmp) != need) return -1;
        size_t k = need;
        for (size_t j = i; j < count; j++) {
            if (bit_sizes[j] == bit_sizes[i]) {
                out[j] = tmp[--k];
                done[j] = 1;
            }
        }
    }
    return 0;
}

uint64_t ora_plain_batching(uint32_t n, int bits)
{
    uint64_t p = 0;
    if (ora_get_primes(2ull * n, bits, 1, &p) != 1) return 0;
    return p;
}

/* SEAL util/globals.cpp default_coeff_modulus_{128,192,256}. The n<=8192 128-bit rows are pinned
 * by logproof/src/rings.rs:36-125; (1024, 192/256) by seal_fhe/src/encryption_parameters.rs:340-365. */
size_t ora_bfv_default(uint32_t n, int sec, uint64_t *out)
{
    static const uint64_t d128_1024[] = {0x7e00001};
    static const uint64_t d128_2048[] = {0x3fffffff000001};
    static const uint64_t d128_4096[] = {0xffffee001, 0xffffc4001, 0x1ffffe0001};
    static const uint64_t d128_8192[] = {0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001};
    static const uint64_t d128_16384[] = {0xfffffffd8001,  0xfffffffa0001,  0xfffffff00001,
                                          0x1fffffff68001, 0x1fffffff50001, 0x1ffffffee8001,
                                          0x1ffffffea0001, 0x1ffffffe88001, 0x1ffffffe48001};
    static const uint64_t d128_32768[] = {0x7fffffffe90001, 0x7fffffffbf0001, 0x7fffffffbd0001, 0x7fffffffba0001,
                                          0x7fffffffaa0001, 0x7fffffffa50001, 0x7fffffff9f0001, 0x7fffffff7e0001,
                                          0x7fffffff770001, 0x7fffffff380001, 0x7fffffff330001, 0x7fffffff2d0001,
                                          0x7fffffff170001, 0x7fffffff150001, 0x7ffffffef00001, 0xfffffffff70001};
    static const uint64_t d192_1024[] = {0x7f001};
    static const uint64_t d256_1024[] = {0x3001};
    const uint64_t *src = NULL;
    size_t cnt = 0;
#define PICK(arr)                             \
    do {                                      \
        src = arr;                            \
        cnt = sizeof(arr) / sizeof(arr[0]);   \
    } while (0)
    if (sec == 128) {
        switch (n) {
        case 1024: PICK(d128_1024); break;
        case 2048: PICK(d128_2048); break;
        case 4096: PICK(d128_4096); break;
        case 8192: PICK(d128_8192); break;
        case 16384: PICK(d128_16384); break;
        case 32768: PICK(d128_32768); break;
        default: break;
        }
    } else if (sec == 192 && n == 1024) {
        PICK(d192_1024);
    } else if (sec == 256 && n == 1024) {
        PICK(d256_1024);
    }
#undef PICK
    if (!src) return 0;
    memcpy(out, src, cnt * sizeof(uint64_t));
    return cnt;
}

uint64_t ora_invmod(uint64_t a, const ora_mod *m)
{
    /* extended Euclid; works for any modulus coprime to a */
    __int128 r0 = (__int128)m->q, r1 = (__int128)(a % m->q), s0 = 0, s1 = 1;
    while (r1 != 0) {
        __int128 qq = r0 / r1;
        __int128 t = r0 - qq * r1;
        r0 = r1;
        r1 = t;
        t = s0 - qq * s1;
        s0 = s1;
        s1 = t;
    }
    if (r0 != 1) return 0;
    if (s0 < 0) s0 += (__int128)m->q;
    return (uint64_t)s0;
}

/* SEAL try_minimal_primitive_root: the smallest primitive 2n-th root of unity mod q.
 * Pinned bit-for-bit by the key fixtures seal_fhe/tests/data/ (SURVEY 8c). */
uint64_t ora_minimal_primitive_root(uint32_t two_n, uint64_t q)
{
    if ((q - 1) % two_n != 0) return 0;
    ora_mod m;
    ora_mod_init(&m, q);
    uint64_t e = (q - 1) / two_n;
    uint64_t root = 0;
    for (uint64_t g = 2; g < 1000; g++) {
        uint64_t c = ora_powmod(g, e, &m);
        if (ora_powmod(c, two_n / 2, &m) == q - 1) {
            root = c;
            break;
        }
    }
    if (!root) return 0;
    uint64_t sq = ora_mulmod(root, root, &m);
    uint64_t cur = root, best = root;
    for (uint32_t i = 0; i < two_n / 2; i++) {
        if (cur < best) best = cur;
        cur = ora_mulmod(cur, sq, &m);
    }
    return best;
}

/* ------------------------------------------------------------------ NTT */

int ora_ntt_init(ora_ntt *t, int logn, uint64_t q)
{
    memset(t, 0, sizeof(*t));
    t->logn = logn;
    t->n = 1u << logn;
    ora_mod_init(&t->mod, q);
    t->root = ora_minimal_primitive_root(2u * t->n, q);
    if (!t->root) return -1;
    uint32_t n = t->n;
    t->rp = (ora_mulop *)malloc(sizeof(ora_mulop) * n);
    t->irp = (ora_mulop *)malloc(sizeof(ora_mulop) * n);
    uint64_t iroot = ora_invmod(t->root, &t->mod);
    uint64_t p = 1, ip = 1;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t k = ora_bitrev(i, logn);
        ora_mulop_init(&t->rp[k], p, &t->mod);
        ora_mulop_init(&t->irp[k], ip, &t->mod);
        p = ora_mulmod(p, t->root, &t->mod);
        ip = ora_mulmod(ip, iroot, &t->mod);
    }
    ora_mulop_init(&t->inv_n, ora_invmod(n, &t->mod), &t->mod);
    return 0;
}

void ora_ntt_free(ora_ntt *t)
{
    free(t->rp);
    free(t->irp);
    t->rp = t->irp = NULL;
}

/* Forward negacyclic NTT, Cooley-Tukey, natural order in -> bit-reversed order out
 * (SEAL ntt_negacyclic_harvey / DWTHandler::transform_to_rev, Harvey lazy butterflies). */
void ora_ntt_fwd(const ora_ntt *t, uint64_t *x)
{
    const uint64_t q = t->mod.q, two_q = q << 1;
    const uint32_t n = t->n;
    uint32_t gap = n >> 1;
    for (uint32_t m = 1; m < n; m <<= 1, gap >>= 1) {
        for (uint32_t i = 0; i < m; i++) {
            const ora_mulop w = t->rp[m + i];
            uint64_t *a = x + 2 * i * gap, *b = a + gap;
            for (uint32_t j = 0; j < gap; j++) {
                uint64_t u = a[j];
                u = u >= two_q ? u - two_q : u;
                uint64_t v = ora_mulop_lazy(b[j], &w, q);
                a[j] = u + v;
                b[j] = u + two_q - v;
            }
        }
    }
    for (uint32_t j = 0; j < n; j++) {
        uint64_t v = x[j];
        v = v >= two_q ? v - two_q : v;
        x[j] = v >= q ? v - q : v;
    }
}

/* Inverse: Gentleman-Sande, bit-reversed in -> natural out, n^{-1} folded at the end
 * (SEAL inverse_ntt_negacyclic_harvey / transform_from_rev). */
void ora_ntt_inv(const ora_ntt *t, uint64_t *x)
{
    const uint64_t q = t->mod.q, two_q = q << 1;
    const uint32_t n = t->n;
    uint32_t gap = 1;
    for (uint32_t m = n >> 1; m >= 1; m >>= 1, gap <<= 1) {
        for (uint32_t i = 0; i < m; i++) {
            const ora_mulop w = t->irp[m + i];
            uint64_t *a = x + 2 * i * gap, *b = a + gap;
            for (uint32_t j = 0; j < gap; j++) {
                uint64_t u = a[j], v = b[j];
                uint64_t s = u + v;
                a[j] = s >= two_q ? s - two_q : s;
                b[j] = ora_mulop_lazy(u + two_q - v, &w, q);
            }
        }
    }
    for (uint32_t j = 0; j < n; j++) x[j] = ora_mulop_mod(x[j], &t->inv_n, q);
}

/* ------------------------------------------------------------------ base converter */

void ora_baseconv_init(ora_baseconv *bc, const ora_mod *in, size_t in_n, const ora_mod *out, size_t out_n)
{
    memset(bc, 0, sizeof(*bc));
    bc->in_n = in_n;
    bc->out_n = out_n;
    memcpy(bc->in, in, sizeof(ora_mod) * in_n);
    memcpy(bc->out, out, sizeof(ora_mod) * out_n);
    for (size_t i = 0; i < in_n; i++) {
        uint64_t p = 1 % in[i].q;
        for (size_t k = 0; k < in_n; k++)
            if (k != i) p = ora_mulmod(p, ora_reduce64(in[k].q, &in[i]), &in[i]);
        ora_mulop_init(&bc->inv_punct[i], ora_invmod(p, &in[i]), &in[i]);
    }
    for (size_t j = 0; j < out_n; j++) {
        for (size_t i = 0; i < in_n; i++) {
            uint64_t p = 1 % out[j].q;
            for (size_t k = 0; k < in_n; k++)
                if (k != i) p = ora_mulmod(p, ora_reduce64(in[k].q, &out[j]), &out[j]);
            bc->matrix[j][i] = p;
        }
    }
}

/* out_j = sum_i [x_i * (Q/q_i)^{-1}]_{q_i} * [Q/q_i]_{p_j}  mod p_j   (no alpha correction). */
void ora_baseconv_apply(const ora_baseconv *bc, const uint64_t *in, uint64_t *out, size_t n)
{
    uint64_t y[ORA_MAXP];
    for (size_t k = 0; k < n; k++) {
        for (size_t i = 0; i < bc->in_n; i++)
            y[i] = ora_mulop_mod(in[i * n + k], &bc->inv_punct[i], bc->in[i].q);
        for (size_t j = 0; j < bc->out_n; j++) {
            u128 acc = 0; /* in_n <= 24 products of < 2^61 * 2^61 bits: no overflow */
            for (size_t i = 0; i < bc->in_n; i++) acc += (u128)y[i] * bc->matrix[j][i];
            out[j * n + k] = ora_reduce128(acc, &bc->out[j]);
        }
    }
}

/* ------------------------------------------------------------------ tiny bigint */

typedef struct {
    uint64_t w[ORA_MAXP + 2];
    size_t len;
} ora_big;

static void big_set(ora_big *b, uint64_t v)
{
    memset(b, 0, sizeof(*b));
    b->w[0] = v;
    b->len = 1;
}

static void big_mul_u64(ora_big *b, uint64_t v)
{
    uint64_t carry = 0;
    for (size_t i = 0; i < b->len; i++) {
        u128 p = (u128)b->w[i] * v + carry;
        b->w[i] = (uint64_t)p;
        carry = (uint64_t)(p >> 64);
    }
    if (carry) b->w[b->len++] = carry;
}

static uint64_t big_divmod_u64(ora_big *b, uint64_t d) /* b /= d, returns remainder */
{
    uint64_t rem = 0;
    for (size_t i = b->len; i-- > 0;) {
        u128 cur = ((u128)rem << 64) | b->w[i];
        b->w[i] = (uint64_t)(cur / d);
        rem = (uint64_t)(cur % d);
    }
    while (b->len > 1 && b->w[b->len - 1] == 0) b->len--;
    return rem;
}

static uint64_t big_mod_u64(const ora_big *b, uint64_t d)
{
    uint64_t rem = 0;
    for (size_t i = b->len; i-- > 0;) {
        u128 cur = ((u128)rem << 64) | b->w[i];
        rem = (uint64_t)(cur % d);
    }
    return rem;
}

static int big_bits(const ora_big *b)
{
    return (int)(64 * (b->len - 1)) + ora_bitcount(b->w[b->len - 1]);
}

/* ------------------------------------------------------------------ context */

static uint64_t prod_mod(const ora_mod *base, size_t cnt, const ora_mod *m)
{
    uint64_t p = 1 % m->q;
    for (size_t i = 0; i < cnt; i++) p = ora_mulmod(p, ora_reduce64(base[i].q, m), m);
    return p;
}

ora_ctx *ora_ctx_create(uint32_t n, const uint64_t *coeff_modulus, size_t count, uint64_t plain_modulus)
{
    if (count == 0 || count > ORA_MAXP - 2 || n < 2 || (n & (n - 1)) || plain_modulus < 2) return NULL;
    ora_ctx *c = (ora_ctx *)calloc(1, sizeof(ora_ctx));
    c->n = n;
    c->logn = ora_bitcount(n) - 1;
    c->key_count = count;
    c->K = count > 1 ? count - 1 : 1;
    for (size_t i = 0; i < count; i++) {
        ora_mod_init(&c->key_mod[i], coeff_modulus[i]);
        if (ora_ntt_init(&c->key_ntt[i], c->logn, coeff_modulus[i]) != 0) {
            ora_ctx_destroy(c);
            return NULL;
        }
    }
    ora_mod_init(&c->t, plain_modulus);
    c->t_batching = 0;
    if (ora_is_prime(plain_modulus) && (plain_modulus - 1) % (2ull * n) == 0) {
        if (ora_ntt_init(&c->t_ntt, c->logn, plain_modulus) == 0) c->t_batching = 1;
    }
    const size_t K = c->K;
    const ora_mod *q = c->key_mod;

    /* q as a big integer: bits, q mod t, floor(q/t) mod q_i */
    ora_big Q;
    big_set(&Q, 1);
    for (size_t i = 0; i < K; i++) big_mul_u64(&Q, q[i].q);
    c->total_coeff_bits = big_bits(&Q);
    c->q_mod_t = big_mod_u64(&Q, plain_modulus);
    ora_big Qdt = Q;
    big_divmod_u64(&Qdt, plain_modulus);
    c->upper_half_threshold = (plain_modulus + 1) >> 1;
    c->fast_plain_lift = 1;
    for (size_t i = 0; i < K; i++) {
        c->coeff_div_plain[i] = big_mod_u64(&Qdt, q[i].q);
        if (plain_modulus >= q[i].q) c->fast_plain_lift = 0;
        c->upper_half_increment[i] = q[i].q - plain_modulus; /* valid only with fast lift */
    }

    /* ---- RNSTool::initialize ---- */
    c->B_size = K;
    if (32 + c->t.bits + c->total_coeff_bits >= 61 * (int)K + 61) c->B_size++;
    c->Bsk_size = c->B_size + 1;
    uint64_t aux[ORA_MAXP + 2];
    size_t need = c->B_size + 2;
    if (ora_get_primes(2ull * n, 61, need, aux) != need) {
        ora_ctx_destroy(c);
        return NULL;
    }
    ora_mod_init(&c->m_sk, aux[0]);
    ora_mod_init(&c->gamma, aux[1]);
    for (size_t j = 0; j < c->B_size; j++) {
        ora_mod_init(&c->B[j], aux[2 + j]);
        c->Bsk[j] = c->B[j];
    }
    c->Bsk[c->B_size] = c->m_sk;
    ora_mod_init(&c->m_tilde, ((uint64_t)1) << 32);
    for (size_t j = 0; j < c->Bsk_size; j++) {
        if (ora_ntt_init(&c->bsk_ntt[j], c->logn, c->Bsk[j].q) != 0) {
            ora_ctx_destroy(c);
            return NULL;
        }
    }
    ora_baseconv_init(&c->q_to_Bsk, q, K, c->Bsk, c->Bsk_size);
    ora_baseconv_init(&c->q_to_mtilde, q, K, &c->m_tilde, 1);
    ora_baseconv_init(&c->B_to_q, c->B, c->B_size, q, K);
    ora_baseconv_init(&c->B_to_msk, c->B, c->B_size, &c->m_sk, 1);
    ora_mod tg[2] = {c->t, c->gamma};
    ora_baseconv_init(&c->q_to_tgamma, q, K, tg, 2);

    for (size_t i = 0; i < K; i++) {
        c->m_tilde_mod_q[i] = ora_reduce64(c->m_tilde.q, &q[i]);
        c->prod_B_mod_q[i] = prod_mod(c->B, c->B_size, &q[i]);
        uint64_t tgq = ora_mulmod(ora_reduce64(plain_modulus, &q[i]), ora_reduce64(c->gamma.q, &q[i]), &q[i]);
        ora_mulop_init(&c->prod_t_gamma_mod_q[i], tgq, &q[i]);
    }
    for (size_t j = 0; j < c->Bsk_size; j++) {
        uint64_t pq = prod_mod(q, K, &c->Bsk[j]);
        c->prod_q_mod_Bsk[j] = pq;
        ora_mulop_init(&c->inv_prod_q_mod_Bsk[j], ora_invmod(pq, &c->Bsk[j]), &c->Bsk[j]);
        ora_mulop_init(&c->inv_mtilde_mod_Bsk[j], ora_invmod(ora_reduce64(c->m_tilde.q, &c->Bsk[j]), &c->Bsk[j]),
                       &c->Bsk[j]);
    }
    ora_mulop_init(&c->inv_prod_B_mod_msk, ora_invmod(prod_mod(c->B, c->B_size, &c->m_sk), &c->m_sk), &c->m_sk);
    {
        uint64_t pq = prod_mod(q, K, &c->m_tilde);
        uint64_t inv = ora_invmod(pq, &c->m_tilde);
        c->neg_inv_prod_q_mod_mtilde = ora_negmod(inv, &c->m_tilde);
    }
    for (int k = 0; k < 2; k++) {
        uint64_t pq = prod_mod(q, K, &tg[k]);
        uint64_t inv = ora_invmod(pq, &tg[k]);
        ora_mulop_init(&c->neg_inv_q_mod_t_gamma[k], ora_negmod(inv, &tg[k]), &tg[k]);
    }
    ora_mulop_init(&c->inv_gamma_mod_t, ora_invmod(ora_reduce64(c->gamma.q, &c->t), &c->t), &c->t);

    if (c->key_count > 1) {
        uint64_t qsp = c->key_mod[c->key_count - 1].q;
        for (size_t i = 0; i < K; i++)
            ora_mulop_init(&c->inv_q_last_mod_q[i], ora_invmod(ora_reduce64(qsp, &q[i]), &q[i]), &q[i]);
    }

    /* BatchEncoder index map (SEAL batchencoder.cpp populate_matrix_reps_index_map;
     * exercised by seal_fhe/src/encoder.rs:75-190) */
    if (c->t_batching) {
        c->batch_index_map = (uint32_t *)malloc(sizeof(uint32_t) * n);
        uint32_t row = n >> 1, m = n << 1;
        uint64_t pos = 1;
        for (uint32_t i = 0; i < row; i++) {
            uint32_t i1 = (uint32_t)((pos - 1) >> 1);
            uint32_t i2 = (uint32_t)((m - pos - 1) >> 1);
            c->batch_index_map[i] = ora_bitrev(i1, c->logn);
            c->batch_index_map[row | i] = ora_bitrev(i2, c->logn);
            pos = (pos * 3) & (m - 1);
        }
    }
    return c;
}

void ora_ctx_destroy(ora_ctx *c)
{
    if (!c) return;
    for (size_t i = 0; i < ORA_MAXP; i++) {
        ora_ntt_free(&c->key_ntt[i]);
        ora_ntt_free(&c->bsk_ntt[i]);
    }
    ora_ntt_free(&c->t_ntt);
    free(c->batch_index_map);
    free(c);
}

uint32_t ora_ctx_n(const ora_ctx *c) { return c->n; }
size_t ora_ctx_K(const ora_ctx *c) { return c->K; }
size_t ora_ctx_key_count(const ora_ctx *c) { return c->key_count; }
uint64_t ora_ctx_prime(const ora_ctx *c, size_t i) { return c->key_mod[i].q; }
uint64_t ora_ctx_plain(const ora_ctx *c) { return c->t.q; }
size_t ora_ctx_bsk_count(const ora_ctx *c) { return c->Bsk_size; }
uint64_t ora_ctx_bsk_prime(const ora_ctx *c, size_t j) { return c->Bsk[j].q; }
uint64_t ora_ctx_gamma(const ora_ctx *c) { return c->gamma.q; }
int ora_ctx_total_coeff_bits(const ora_ctx *c) { return c->total_coeff_bits; }

void ora_ntt_forward(const ora_ctx *c, size_t idx, uint64_t *x) { ora_ntt_fwd(&c->key_ntt[idx], x); }
void ora_ntt_inverse(const ora_ctx *c, size_t idx, uint64_t *x) { ora_ntt_inv(&c->key_ntt[idx], x); }
void ora_ntt_forward_bsk(const ora_ctx *c, size_t j, uint64_t *x) { ora_ntt_fwd(&c->bsk_ntt[j], x); }
void ora_ntt_inverse_bsk(const ora_ctx *c, size_t j, uint64_t *x) { ora_ntt_inv(&c->bsk_ntt[j], x); }

int ora_ntt_forward_plain(const ora_ctx *c, uint64_t *x)
{
    if (!c->t_batching) return -1;
    ora_ntt_fwd(&c->t_ntt, x);
    return 0;
}

int ora_ntt_inverse_plain(const ora_ctx *c, uint64_t *x)
{
    if (!c->t_batching) return -1;
    ora_ntt_inv(&c->t_ntt, x);
    return 0;
}
