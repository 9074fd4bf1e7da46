/*
 * oracle/ora_eval.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the evaluator operations on the hot path:
 *   multiply     <- seal_fhe/src/evaluator_base.rs:184-212   (Evaluator_Multiply  -> SEAL bfv_multiply, BEHZ)
 *   relinearize  <- seal_fhe/src/bfv_evaluator.rs:143-175    (Evaluator_Relinearize -> switch_key_inplace)
 *   rotate_*     <- seal_fhe/src/bfv_evaluator.rs:177-247    (Evaluator_RotateRows/Columns -> apply_galois + switch_key)
 *   multiply_plain, add/sub/negate, add_plain/sub_plain <- seal_fhe/src/evaluator_base.rs:89-123,166-182,320-404
 * Algorithms restate SEAL 4.0 evaluator.cpp / util/rns.cpp / util/galois.cpp / util/scalingvariant.cpp
 * (source absent from /root/reference; see bfv_oracle.h for the pinning status).
 */
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "ora_internal.h"

static uint64_t *poly_alloc(size_t words)
{
    return (uint64_t *)malloc(words * sizeof(uint64_t));
}

/* ------------------------------------------------------------------ add / sub / negate */

/* SEAL is built by seal_fhe with SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT (seal_fhe/build.rs:46-66): every Evaluator operation
 * ends with `if (result.is_transparent()) throw logic_error("result ciphertext is transparent")` [RECALLED: SEAL 4.0
 * evaluator.cpp -- negate, add, sub, multiply, square, relinearize_internal, mod_switch_*_to_next, add_plain, sub_plain,
 * multiply_plain, apply_galois]; Ciphertext::is_transparent = all polynomials from index 1 on are zero.  Pinned by
 * sunscreen/tests/features.rs:8-34 (`a * 0` fails without the transparent-ciphertexts feature). */
static int is_transparent(const ora_ctx *c, const uint64_t *ct, size_t s)
{
    const size_t K = c->K, n = c->n;
    for (size_t k = K * n; k < s * K * n; k++)
        if (ct[k]) return 0;
    return 1;
}

static int addsub(const ora_ctx *c, const uint64_t *a, size_t sa, const uint64_t *b, size_t sb, uint64_t *out, int sub)
{
    if (sa < 2 || sb < 2) return ORA_E_INVALIDARG;
    const size_t K = c->K, n = c->n;
    size_t smax = sa > sb ? sa : sb, smin = sa < sb ? sa : sb;
    for (size_t p = 0; p < smax; p++) {
        for (size_t i = 0; i < K; i++) {
            const ora_mod *m = &c->key_mod[i];
            const size_t off = (p * K + i) * n;
            for (size_t k = 0; k < n; k++) {
                uint64_t r;
                if (p < smin)
                    r = sub ? ora_submod(a[off + k], b[off + k], m) : ora_addmod(a[off + k], b[off + k], m);
                else if (sa > sb)
                    r = a[off + k];
                else
                    r = sub ? ora_negmod(b[off + k], m) : b[off + k];
                out[off + k] = r;
            }
        }
    }
    return 0;
}

int ora_add(const ora_ctx *c, const uint64_t *a, size_t sa, const uint64_t *b, size_t sb, uint64_t *out)
{
    int rc = addsub(c, a, sa, b, sb, out, 0);
    if (!rc && is_transparent(c, out, sa > sb ? sa : sb)) return ORA_E_TRANSPARENT;
    return rc;
}

int ora_sub(const ora_ctx *c, const uint64_t *a, size_t sa, const uint64_t *b, size_t sb, uint64_t *out)
{
    int rc = addsub(c, a, sa, b, sb, out, 1);
    if (!rc && is_transparent(c, out, sa > sb ? sa : sb)) return ORA_E_TRANSPARENT;
    return rc;
}

int ora_negate(const ora_ctx *c, const uint64_t *a, size_t sa, uint64_t *out)
{
    if (sa < 2) return ORA_E_INVALIDARG;
    const size_t K = c->K, n = c->n;
    for (size_t p = 0; p < sa; p++)
        for (size_t i = 0; i < K; i++)
            for (size_t k = 0; k < n; k++) out[(p * K + i) * n + k] = ora_negmod(a[(p * K + i) * n + k], &c->key_mod[i]);
    return is_transparent(c, out, sa) ? ORA_E_TRANSPARENT : 0;
}

/* ------------------------------------------------------------------ BEHZ pieces (SEAL util/rns.cpp) */

/* fastbconv_m_tilde followed by sm_mrq: base q -> base Bsk with the q-overflow removed. */
void ora_behz_extend(const ora_ctx *c, const uint64_t *in_q, uint64_t *out_bsk)
{
    const size_t K = c->K, n = c->n, S = c->Bsk_size;
    uint64_t *tmp = poly_alloc(K * n);
    uint64_t *ext = poly_alloc((S + 1) * n);
    /* x * m_tilde mod q_i */
    for (size_t i = 0; i < K; i++)
        for (size_t k = 0; k < n; k++) tmp[i * n + k] = ora_mulmod(in_q[i * n + k], c->m_tilde_mod_q[i], &c->key_mod[i]);
    ora_baseconv_apply(&c->q_to_Bsk, tmp, ext, n);
    ora_baseconv_apply(&c->q_to_mtilde, tmp, ext + S * n, n);
    /* sm_mrq */
    const uint64_t mt = c->m_tilde.q, mt_half = mt >> 1;
    const uint64_t *xm = ext + S * n;
    for (size_t j = 0; j < S; j++) {
        const ora_mod *p = &c->Bsk[j];
        for (size_t k = 0; k < n; k++) {
            uint64_t r = ora_mulmod(xm[k], c->neg_inv_prod_q_mod_mtilde, &c->m_tilde);
            if (r >= mt_half) r += p->q - mt; /* centred representative of r modulo p */
            uint64_t v = ora_mulmod_add(r, c->prod_q_mod_Bsk[j], ext[j * n + k], p);
            out_bsk[j * n + k] = ora_mulop_mod(v, &c->inv_mtilde_mod_Bsk[j], p->q);
        }
    }
    free(tmp);
    free(ext);
}

/* fast_floor (q U Bsk -> Bsk) then fastbconv_sk (Bsk -> q). */
void ora_behz_floor_sk(const ora_ctx *c, const uint64_t *in, uint64_t *out_q)
{
    const size_t K = c->K, n = c->n, S = c->Bsk_size, Bn = c->B_size;
    uint64_t *fl = poly_alloc(S * n);
    ora_baseconv_apply(&c->q_to_Bsk, in, fl, n);
    const uint64_t *in_bsk = in + K * n;
    for (size_t j = 0; j < S; j++) {
        const ora_mod *p = &c->Bsk[j];
        for (size_t k = 0; k < n; k++) {
            uint64_t d = in_bsk[j * n + k] + (p->q - fl[j * n + k]);
            fl[j * n + k] = ora_mulop_mod(d, &c->inv_prod_q_mod_Bsk[j], p->q);
        }
    }
    /* Shenoy-Kumaresan */
    uint64_t *alpha = poly_alloc(n);
    ora_baseconv_apply(&c->B_to_q, fl, out_q, n);
    ora_baseconv_apply(&c->B_to_msk, fl, alpha, n);
    const uint64_t msk = c->m_sk.q, msk_half = msk >> 1;
    for (size_t k = 0; k < n; k++) {
        uint64_t d = alpha[k] + (msk - fl[Bn * n + k]);
        alpha[k] = ora_mulop_mod(d, &c->inv_prod_B_mod_msk, msk);
    }
    for (size_t i = 0; i < K; i++) {
        const ora_mod *q = &c->key_mod[i];
        const uint64_t pb = c->prod_B_mod_q[i], npb = q->q - pb;
        for (size_t k = 0; k < n; k++) {
            uint64_t a = alpha[k];
            if (a > msk_half)
                out_q[i * n + k] = ora_mulmod_add(msk - a, pb, out_q[i * n + k], q);
            else
                out_q[i * n + k] = ora_mulmod_add(a, npb, out_q[i * n + k], q);
        }
    }
    free(alpha);
    free(fl);
}

/* ------------------------------------------------------------------ multiply (BEHZ) */

int ora_multiply(const ora_ctx *c, const uint64_t *a, size_t sa, const uint64_t *b, size_t sb, uint64_t *out)
{
    if (sa < 2 || sb < 2) return ORA_E_INVALIDARG;
    const size_t K = c->K, n = c->n, S = c->Bsk_size;
    const size_t sd = sa + sb - 1;
    const size_t R = K + S; /* residues per extended poly: q then Bsk */
    uint64_t *ea = poly_alloc(sa * R * n), *eb = poly_alloc(sb * R * n);
    uint64_t *ed = poly_alloc(sd * R * n);
    /* steps (1)-(3): extend each input polynomial to q U Bsk and move to NTT form */
    for (int which = 0; which < 2; which++) {
        const uint64_t *src = which ? b : a;
        uint64_t *dst = which ? eb : ea;
        size_t s = which ? sb : sa;
        for (size_t p = 0; p < s; p++) {
            uint64_t *d = dst + p * R * n;
            memcpy(d, src + p * K * n, K * n * sizeof(uint64_t));
            ora_behz_extend(c, src + p * K * n, d + K * n);
            for (size_t i = 0; i < K; i++) ora_ntt_fwd(&c->key_ntt[i], d + i * n);
            for (size_t j = 0; j < S; j++) ora_ntt_fwd(&c->bsk_ntt[j], d + (K + j) * n);
        }
    }
    /* step (4): dyadic tensor product, per residue */
    memset(ed, 0, sd * R * n * sizeof(uint64_t));
    for (size_t r = 0; r < R; r++) {
        const ora_mod *m = r < K ? &c->key_mod[r] : &c->Bsk[r - K];
        for (size_t pa = 0; pa < sa; pa++) {
            for (size_t pb = 0; pb < sb; pb++) {
                const uint64_t *x = ea + (pa * R + r) * n, *y = eb + (pb * R + r) * n;
                uint64_t *z = ed + ((pa + pb) * R + r) * n;
                for (size_t k = 0; k < n; k++) z[k] = ora_addmod(z[k], ora_mulmod(x[k], y[k], m), m);
            }
        }
    }
    /* steps (5)-(8) */
    const uint64_t t = c->t.q;
    for (size_t p = 0; p < sd; p++) {
        uint64_t *d = ed + p * R * n;
        for (size_t i = 0; i < K; i++) {
            ora_ntt_inv(&c->key_ntt[i], d + i * n);
            for (size_t k = 0; k < n; k++) d[i * n + k] = ora_mulmod(d[i * n + k], t, &c->key_mod[i]);
        }
        for (size_t j = 0; j < S; j++) {
            ora_ntt_inv(&c->bsk_ntt[j], d + (K + j) * n);
            for (size_t k = 0; k < n; k++) d[(K + j) * n + k] = ora_mulmod(d[(K + j) * n + k], t, &c->Bsk[j]);
        }
        ora_behz_floor_sk(c, d, out + p * K * n);
    }
    free(ea);
    free(eb);
    free(ed);
    return is_transparent(c, out, sd) ? ORA_E_TRANSPARENT : 0;
}

/* ------------------------------------------------------------------ key switching */

/* SEAL Evaluator::switch_key_inplace (BFV branch): ct2 += modDown( sum_J NTT(target_J) (.) key[J] ). */
int ora_switch_key(const ora_ctx *c, uint64_t *ct2, const uint64_t *target, const uint64_t *key)
{
    if (c->key_count < 2) return ORA_E_INVALIDARG;
    const size_t K = c->K, n = c->n, KK = c->key_count;
    uint64_t *acc = poly_alloc(2 * KK * n); /* [comp][I][n] */
    uint64_t *tn = poly_alloc(n);
    u128 *lazy = (u128 *)malloc(sizeof(u128) * 2 * n);
    for (size_t I = 0; I < KK; I++) {
        const ora_mod *mI = &c->key_mod[I];
        memset(lazy, 0, sizeof(u128) * 2 * n);
        for (size_t J = 0; J < K; J++) {
            const uint64_t *tj = target + J * n;
            if (c->key_mod[J].q <= mI->q)
                memcpy(tn, tj, n * sizeof(uint64_t));
            else
                for (size_t k = 0; k < n; k++) tn[k] = ora_reduce64(tj[k], mI);
            ora_ntt_fwd(&c->key_ntt[I], tn);
            for (size_t comp = 0; comp < 2; comp++) {
                const uint64_t *kp = key + ((J * 2 + comp) * KK + I) * n;
                u128 *lz = lazy + comp * n;
                for (size_t k = 0; k < n; k++) lz[k] += (u128)tn[k] * kp[k];
            }
            /* K <= 23 summands of < 2^122: no 128-bit overflow, single final reduction */
        }
        for (size_t comp = 0; comp < 2; comp++)
            for (size_t k = 0; k < n; k++) acc[(comp * KK + I) * n + k] = ora_reduce128(lazy[comp * n + k], mI);
    }
    /* modulus switching with rounding by the special prime */
    const ora_mod *msp = &c->key_mod[KK - 1];
    const uint64_t qk = msp->q, qk_half = qk >> 1;
    for (size_t comp = 0; comp < 2; comp++) {
        uint64_t *tl = acc + (comp * KK + (KK - 1)) * n;
        ora_ntt_inv(&c->key_ntt[KK - 1], tl);
        for (size_t k = 0; k < n; k++) tl[k] = ora_reduce64(tl[k] + qk_half, msp);
        for (size_t J = 0; J < K; J++) {
            const ora_mod *mj = &c->key_mod[J];
            const uint64_t half_mod = ora_reduce64(qk_half, mj);
            uint64_t *aj = acc + (comp * KK + J) * n;
            ora_ntt_inv(&c->key_ntt[J], aj);
            uint64_t *dst = ct2 + (comp * K + J) * n;
            for (size_t k = 0; k < n; k++) {
                uint64_t tk = qk > mj->q ? ora_reduce64(tl[k], mj) : tl[k];
                tk = ora_submod(tk, half_mod, mj);
                uint64_t d = ora_submod(aj[k], tk, mj);
                d = ora_mulop_mod(d, &c->inv_q_last_mod_q[J], mj->q);
                dst[k] = ora_addmod(dst[k], d, mj);
            }
        }
    }
    free(lazy);
    free(tn);
    free(acc);
    return 0;
}

int ora_relinearize(const ora_ctx *c, const uint64_t *ct3, const uint64_t *rk, uint64_t *out2)
{
    const size_t K = c->K, n = c->n;
    uint64_t *target = poly_alloc(K * n);
    memcpy(target, ct3 + 2 * K * n, K * n * sizeof(uint64_t));
    memmove(out2, ct3, 2 * K * n * sizeof(uint64_t));
    int rc = ora_switch_key(c, out2, target, rk);
    free(target);
    if (!rc && is_transparent(c, out2, 2)) return ORA_E_TRANSPARENT;
    return rc;
}

/* ------------------------------------------------------------------ Galois */

/* SEAL GaloisTool::apply_galois (coefficient form). */
int ora_apply_galois_poly(const ora_ctx *c, const uint64_t *in, uint32_t elt, uint64_t *out)
{
    const size_t K = c->K;
    const uint32_t n = c->n;
    if (!(elt & 1) || elt >= 2 * n) return ORA_E_INVALIDARG;
    for (size_t i = 0; i < K; i++) {
        const uint64_t q = c->key_mod[i].q;
        const uint64_t *x = in + i * n;
        uint64_t *y = out + i * n;
        uint64_t raw = 0;
        for (uint32_t k = 0; k < n; k++, raw += elt) {
            uint32_t idx = (uint32_t)(raw & (n - 1));
            uint64_t v = x[k];
            if ((raw >> c->logn) & 1) v = v ? q - v : 0;
            y[idx] = v;
        }
    }
    return 0;
}

/* SEAL GaloisTool::get_elt_from_step */
uint32_t ora_galois_elt_from_step(const ora_ctx *c, int step)
{
    const uint32_t n = c->n, m = 2 * n;
    if (step == 0) return m - 1;
    uint32_t pos = (uint32_t)(step < 0 ? -step : step);
    if (pos >= (n >> 1)) return 0;
    uint32_t s = step < 0 ? (n >> 1) - pos : pos;
    uint64_t e = 1;
    for (uint32_t i = 0; i < s; i++) e = (e * 3) & (m - 1);
    return (uint32_t)e;
}

int ora_apply_galois(const ora_ctx *c, const uint64_t *ct2, uint32_t elt, const uint64_t *const *gk, uint64_t *out2)
{
    const size_t K = c->K, n = c->n;
    if (!(elt & 1) || elt >= 2 * n) return ORA_E_INVALIDARG;
    const uint64_t *key = gk[(elt - 1) >> 1];
    if (!key) return ORA_E_NOKEY;
    uint64_t *res = poly_alloc(2 * K * n);
    uint64_t *target = poly_alloc(K * n);
    ora_apply_galois_poly(c, ct2, elt, res);
    ora_apply_galois_poly(c, ct2 + K * n, elt, target);
    memset(res + K * n, 0, K * n * sizeof(uint64_t));
    int rc = ora_switch_key(c, res, target, key);
    memcpy(out2, res, 2 * K * n * sizeof(uint64_t));
    free(res);
    free(target);
    if (!rc && is_transparent(c, out2, 2)) return ORA_E_TRANSPARENT;
    return rc;
}

static int rotate_internal(const ora_ctx *c, uint64_t *ct2, int steps, const uint64_t *const *gk)
{
    if (steps == 0) return 0;
    const size_t K = c->K, n = c->n;
    uint32_t elt = ora_galois_elt_from_step(c, steps);
    if (!elt) return ORA_E_INVALIDARG;
    if (gk[(elt - 1) >> 1]) return ora_apply_galois(c, ct2, elt, gk, ct2);
    /* SEAL util::naf(): non-adjacent form, low bit first */
    int v = steps < 0 ? -steps : steps, sign = steps < 0;
    int parts[40], np = 0;
    for (int i = 0; v; i++) {
        int zi = (v & 1) ? 2 - (v & 3) : 0;
        v = (v - zi) >> 1;
        if (zi) parts[np++] = (sign ? -zi : zi) * (1 << i);
    }
    if (np == 1) return ORA_E_NOKEY;
    for (int i = 0; i < np; i++) {
        int s = parts[i];
        if ((uint32_t)(s < 0 ? -s : s) != (n >> 1)) {
            int rc = rotate_internal(c, ct2, s, gk);
            if (rc) return rc;
        }
    }
    (void)K;
    return 0;
}

int ora_rotate_rows(const ora_ctx *c, const uint64_t *ct2, int steps, const uint64_t *const *gk, uint64_t *out2)
{
    if (!c->t_batching) return ORA_E_INVALIDARG;
    memmove(out2, ct2, 2 * c->K * c->n * sizeof(uint64_t));
    return rotate_internal(c, out2, steps, gk);
}

int ora_rotate_columns(const ora_ctx *c, const uint64_t *ct2, const uint64_t *const *gk, uint64_t *out2)
{
    if (!c->t_batching) return ORA_E_INVALIDARG;
    return ora_apply_galois(c, ct2, 2 * c->n - 1, gk, out2);
}

/* ------------------------------------------------------------------ plaintext ops */

/* SEAL util::multiply_add_plain_with_scaling_variant / multiply_sub_...: c0 +/-= round(q*m/t) */
static int plain_addsub(const ora_ctx *c, const uint64_t *ct, size_t s, const uint64_t *plain, size_t pc, uint64_t *out,
                        int sub)
{
    const size_t K = c->K, n = c->n;
    if (s < 2 || pc > n) return ORA_E_INVALIDARG;
    if (out != ct) memmove(out, ct, s * K * n * sizeof(uint64_t));
    const uint64_t t = c->t.q;
    for (size_t k = 0; k < pc; k++) {
        if (plain[k] >= t) return ORA_E_INVALIDARG;
        u128 num = (u128)plain[k] * c->q_mod_t + c->upper_half_threshold;
        uint64_t fix = (uint64_t)(num / t);
        for (size_t i = 0; i < K; i++) {
            const ora_mod *m = &c->key_mod[i];
            uint64_t v = ora_mulmod_add(plain[k], c->coeff_div_plain[i], fix, m);
            uint64_t *d = out + i * n + k;
            *d = sub ? ora_submod(*d, v, m) : ora_addmod(*d, v, m);
        }
    }
    return 0;
}

int ora_add_plain(const ora_ctx *c, const uint64_t *ct, size_t s, const uint64_t *plain, size_t pc, uint64_t *out)
{
    int rc = plain_addsub(c, ct, s, plain, pc, out, 0);
    if (!rc && is_transparent(c, out, s)) return ORA_E_TRANSPARENT;
    return rc;
}

int ora_sub_plain(const ora_ctx *c, const uint64_t *ct, size_t s, const uint64_t *plain, size_t pc, uint64_t *out)
{
    int rc = plain_addsub(c, ct, s, plain, pc, out, 1);
    if (!rc && is_transparent(c, out, s)) return ORA_E_TRANSPARENT;
    return rc;
}


/* Lift one plaintext coefficient (mod t) to q_i: values >= (t+1)/2 represent negatives, i.e. c - t.
 * With SEAL's fast plain lift (t < every q_i) this is c + (q_i - t); otherwise SEAL adds the
 * multi-precision q - t and decomposes, which is the same residue (c - t) mod q_i. */
static inline uint64_t lift_plain_coeff(const ora_ctx *c, uint64_t v, size_t i)
{
    const ora_mod *m = &c->key_mod[i];
    if (v < c->upper_half_threshold) return ora_reduce64(v, m);
    return ora_negmod(ora_reduce64(c->t.q - v, m), m);
}

/* SEAL Evaluator::multiply_plain_normal. */
int ora_multiply_plain(const ora_ctx *c, const uint64_t *ct, size_t s, const uint64_t *plain, size_t pc, uint64_t *out)
{
    const size_t K = c->K, n = c->n;
    if (s < 2 || pc > n) return ORA_E_INVALIDARG;
    size_t nonzero = 0, sig = 0;
    for (size_t k = 0; k < pc; k++) {
        if (plain[k] >= c->t.q) return ORA_E_INVALIDARG;
        if (plain[k]) {
            nonzero++;
            sig = k + 1;
        }
    }
    if (out != ct) memmove(out, ct, s * K * n * sizeof(uint64_t));
    if (nonzero == 1) {
        /* monomial: negacyclic shift by mono_exponent and scalar multiply; with fast plain lift the
         * coefficient is used as-is even when it is >= (t+1)/2 (no q_i - t adjustment), otherwise
         * it is lifted like any other coefficient. */
        const size_t e = sig - 1;
        uint64_t *tmp = poly_alloc(n);
        for (size_t p = 0; p < s; p++) {
            for (size_t i = 0; i < K; i++) {
                const ora_mod *m = &c->key_mod[i];
                const uint64_t coeff = c->fast_plain_lift ? plain[e] : lift_plain_coeff(c, plain[e], i);
                uint64_t *x = out + (p * K + i) * n;
                for (size_t k = 0; k < n; k++) {
                    uint64_t v = ora_mulmod(x[k], coeff, m);
                    size_t idx = k + e;
                    if (idx >= n) {
                        idx -= n;
                        v = ora_negmod(v, m);
                    }
                    tmp[idx] = v;
                }
                memcpy(x, tmp, n * sizeof(uint64_t));
            }
        }
        free(tmp);
    } else {
        uint64_t *pl = poly_alloc(n);
        for (size_t i = 0; i < K; i++) {
            const ora_mod *m = &c->key_mod[i];
            memset(pl, 0, n * sizeof(uint64_t));
            for (size_t k = 0; k < pc; k++) pl[k] = lift_plain_coeff(c, plain[k], i);
            ora_ntt_fwd(&c->key_ntt[i], pl);
            for (size_t p = 0; p < s; p++) {
                uint64_t *x = out + (p * K + i) * n;
                ora_ntt_fwd(&c->key_ntt[i], x);
                for (size_t k = 0; k < n; k++) x[k] = ora_mulmod(x[k], pl[k], m);
                ora_ntt_inv(&c->key_ntt[i], x);
            }
        }
        free(pl);
    }
    if (is_transparent(c, out, s)) return ORA_E_TRANSPARENT;
    return 0;
}

/* ------------------------------------------------------------------ cpu_baseline timing legs */

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double ora_bench_mul_relin(const ora_ctx *c, const uint64_t *a, const uint64_t *b, const uint64_t *rk, uint64_t *out,
                           size_t count, int threads)
{
    const size_t K = c->K, n = c->n, ctw = 2 * K * n;
    (void)threads;
    double t0 = now_s();
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
#endif
    for (long i = 0; i < (long)count; i++) {
        uint64_t *tmp = poly_alloc(3 * K * n);
        ora_multiply(c, a + (size_t)i * ctw, 2, b + (size_t)i * ctw, 2, tmp);
        ora_relinearize(c, tmp, rk, out + (size_t)i * ctw);
        free(tmp);
    }
    return now_s() - t0;
}

double ora_bench_ntt(const ora_ctx *c, uint64_t *x, size_t count, size_t nprimes, int threads)
{
    const size_t n = c->n;
    (void)threads;
    double t0 = now_s();
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (long i = 0; i < (long)count; i++) {
        size_t idx = (size_t)i % nprimes;
        ora_ntt_fwd(&c->key_ntt[idx], x + (size_t)i * n);
        ora_ntt_inv(&c->key_ntt[idx], x + (size_t)i * n);
    }
    return now_s() - t0;
}

/* SEAL Evaluator::mod_switch_to_next for BFV (seal_fhe/src/evaluator.rs:84-157): every polynomial is divided-and-rounded by
 * the LAST DATA prime (RNSTool::divide_and_round_q_last_inplace at the data level) and keeps the remaining K-1 residues.
 * out: uint64[s][K-1][n]. */
int ora_mod_switch_to_next(const ora_ctx *c, const uint64_t *ct, size_t s, uint64_t *out)
{
    const size_t n = c->n, K = c->K;
    if (K < 2) return ORA_E_INVALIDARG;
    const ora_mod *ml = &c->key_mod[K - 1];
    const uint64_t half = ml->q >> 1;
    for (size_t p = 0; p < s; p++) {
        const uint64_t *x = ct + p * K * n;
        for (size_t i = 0; i + 1 < K; i++) {
            const ora_mod *m = &c->key_mod[i];
            const uint64_t half_mod = ora_reduce64(half, m);
            const uint64_t inv = ora_invmod(ora_reduce64(ml->q, m), m);
            for (size_t k = 0; k < n; k++) {
                uint64_t last = ora_addmod(x[(K - 1) * n + k], half, ml);
                uint64_t tk = ora_submod(ora_reduce64(last, m), half_mod, m);
                uint64_t d = ora_submod(x[i * n + k], tk, m);
                out[(p * (K - 1) + i) * n + k] = ora_mulmod(d, inv, m);
            }
        }
    }
    for (size_t k = (K - 1) * n; k < s * (K - 1) * n; k++)
        if (out[k]) return 0;
    return ORA_E_TRANSPARENT; /* is_transparent() at the next level's K - 1 residues */
}
