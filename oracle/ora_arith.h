/*
 * oracle/ora_arith.h -- TEST INFRASTRUCTURE ONLY (CPU oracle, see oracle/README.md).
 *
 * Word-level modular arithmetic used by the CPU restatement of the BFV hot path that
 * sits behind seal_fhe::Evaluator (reference boundary: seal_fhe/src/evaluator.rs:7-280).
 * The arithmetic itself lives in Microsoft SEAL 4.0 (Sunscreen fork), an un-vendored
 * submodule that is ABSENT from /root/reference (seal_fhe/SEAL/ is empty), so these are
 * restatements of SEAL's *published* algorithms (Barrett reduction with a precomputed
 * floor(2^128/q), Shoup/Harvey multiplication with a precomputed floor(w*2^64/q)).
 * Every function returns canonical residues in [0,q), which is all that is observable.
 *
 * Nothing under sunscreen_amd/ may include or link this file.
 */
#ifndef ORA_ARITH_H
#define ORA_ARITH_H

#include <stdint.h>
#include <stddef.h>

typedef unsigned __int128 u128;

typedef struct {
    uint64_t q;        /* modulus value (prime, or 2^32 for m_tilde)            */
    uint64_t ratio[2]; /* floor(2^128 / q): [0] low word, [1] high word          */
    int bits;          /* significant bit count                                  */
} ora_mod;

typedef struct {
    uint64_t w;  /* operand in [0,q)                  */
    uint64_t wq; /* floor(w * 2^64 / q)  (Shoup quot.) */
} ora_mulop;

static inline int ora_bitcount(uint64_t v)
{
    return v ? 64 - __builtin_clzll(v) : 0;
}

static inline void ora_mod_init(ora_mod *m, uint64_t q)
{
    m->q = q;
    m->bits = ora_bitcount(q);
    /* floor(2^128 / q) = floor((2^128 - 1) / q) unless q | 2^128 (q a power of two) */
    u128 all = ~(u128)0;
    u128 r = all / q;
    if ((q & (q - 1)) == 0) {
        r = all / q + ((all % q) == (u128)(q - 1) ? 1 : 0);
    }
    m->ratio[0] = (uint64_t)r;
    m->ratio[1] = (uint64_t)(r >> 64);
}

static inline uint64_t ora_mulhi(uint64_t a, uint64_t b)
{
    return (uint64_t)(((u128)a * b) >> 64);
}

/* x mod q for arbitrary 64-bit x (one-word Barrett). */
static inline uint64_t ora_reduce64(uint64_t x, const ora_mod *m)
{
    uint64_t t = ora_mulhi(x, m->ratio[1]);
    uint64_t r = x - t * m->q;
    return r >= m->q ? r - m->q : r;
}

/* x mod q for a 128-bit x (two-word Barrett). */
static inline uint64_t ora_reduce128(u128 x, const ora_mod *m)
{
    uint64_t x0 = (uint64_t)x, x1 = (uint64_t)(x >> 64);
    /* quotient estimate = floor(x * ratio / 2^128), only its low word is needed */
    uint64_t carry = ora_mulhi(x0, m->ratio[0]);
    u128 t2 = (u128)x0 * m->ratio[1];
    uint64_t t1 = (uint64_t)t2 + carry;
    uint64_t t3 = (uint64_t)(t2 >> 64) + (t1 < (uint64_t)t2);
    u128 t4 = (u128)x1 * m->ratio[0];
    uint64_t t5 = t1 + (uint64_t)t4;
    carry = (uint64_t)(t4 >> 64) + (t5 < t1);
    uint64_t qhat = x1 * m->ratio[1] + t3 + carry;
    uint64_t r = x0 - qhat * m->q;
    return r >= m->q ? r - m->q : r;
}

static inline uint64_t ora_mulmod(uint64_t a, uint64_t b, const ora_mod *m)
{
    return ora_reduce128((u128)a * b, m);
}

/* (a*b + c) mod q */
static inline uint64_t ora_mulmod_add(uint64_t a, uint64_t b, uint64_t c, const ora_mod *m)
{
    return ora_reduce128((u128)a * b + c, m);
}

static inline uint64_t ora_addmod(uint64_t a, uint64_t b, const ora_mod *m)
{
    uint64_t s = a + b; /* a,b < q < 2^63 */
    return s >= m->q ? s - m->q : s;
}

static inline uint64_t ora_submod(uint64_t a, uint64_t b, const ora_mod *m)
{
    return a >= b ? a - b : a + m->q - b;
}

static inline uint64_t ora_negmod(uint64_t a, const ora_mod *m)
{
    return a ? m->q - a : 0;
}

static inline void ora_mulop_init(ora_mulop *o, uint64_t w, const ora_mod *m)
{
    o->w = w;
    o->wq = (uint64_t)((((u128)w) << 64) / m->q);
}

/* x*w mod q in [0,2q) for arbitrary 64-bit x (Harvey lazy form). */
static inline uint64_t ora_mulop_lazy(uint64_t x, const ora_mulop *o, uint64_t q)
{
    uint64_t h = ora_mulhi(x, o->wq);
    return x * o->w - h * q;
}

static inline uint64_t ora_mulop_mod(uint64_t x, const ora_mulop *o, uint64_t q)
{
    uint64_t r = ora_mulop_lazy(x, o, q);
    return r >= q ? r - q : r;
}

static inline uint64_t ora_powmod(uint64_t b, uint64_t e, const ora_mod *m)
{
    uint64_t r = 1 % m->q;
    b = ora_reduce64(b, m);
    while (e) {
        if (e & 1) r = ora_mulmod(r, b, m);
        b = ora_mulmod(b, b, m);
        e >>= 1;
    }
    return r;
}

static inline uint32_t ora_bitrev(uint32_t v, int bits)
{
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) {
        r = (r << 1) | ((v >> i) & 1);
    }
    return r;
}

#endif /* ORA_ARITH_H */
