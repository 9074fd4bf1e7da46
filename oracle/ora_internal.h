/* oracle/ora_internal.h -- TEST INFRASTRUCTURE ONLY. Internal layout of the oracle context. */
#ifndef ORA_INTERNAL_H
#define ORA_INTERNAL_H

#include "bfv_oracle.h"
#include "ora_arith.h"

#define ORA_MAXP 24 /* max primes in any base (n=32768 default has 16 key primes; Bsk <= 17) */

typedef struct {
    int logn;
    uint32_t n;
    ora_mod mod;
    uint64_t root;      /* minimal primitive 2n-th root psi                     */
    ora_mulop *rp;      /* rp[k]  = psi^{bitrev(k)}       (k = 1..n-1, rp[0]=1)  */
    ora_mulop *irp;     /* irp[k] = psi^{-bitrev(k)}                             */
    ora_mulop inv_n;    /* n^{-1}                                                */
} ora_ntt;

/* fast base conversion ibase -> obase (SEAL BaseConverter::fast_convert_array) */
typedef struct {
    size_t in_n, out_n;
    ora_mod in[ORA_MAXP], out[ORA_MAXP];
    ora_mulop inv_punct[ORA_MAXP];          /* (prod(ibase)/ibase_i)^{-1} mod ibase_i */
    uint64_t matrix[ORA_MAXP][ORA_MAXP];    /* [j][i] = prod(ibase)/ibase_i mod obase_j */
} ora_baseconv;

struct ora_ctx {
    uint32_t n;
    int logn;
    size_t key_count; /* key-level primes                                   */
    size_t K;         /* data-level primes (= key_count-1, or 1 if single)  */
    ora_mod key_mod[ORA_MAXP];
    ora_ntt key_ntt[ORA_MAXP];
    ora_mod t;
    int t_batching;
    ora_ntt t_ntt;
    int total_coeff_bits; /* bits of prod(data-level q) */

    /* RNS tool for the first data level (SEAL util/rns.cpp RNSTool::initialize) */
    size_t B_size, Bsk_size;
    ora_mod B[ORA_MAXP], Bsk[ORA_MAXP]; /* Bsk = B || m_sk */
    ora_mod m_sk, gamma, m_tilde;
    ora_ntt bsk_ntt[ORA_MAXP];
    ora_baseconv q_to_Bsk, q_to_mtilde, B_to_q, B_to_msk, q_to_tgamma;
    uint64_t m_tilde_mod_q[ORA_MAXP];          /* m_tilde mod q_i                     */
    uint64_t prod_B_mod_q[ORA_MAXP];
    ora_mulop inv_prod_q_mod_Bsk[ORA_MAXP];
    ora_mulop inv_prod_B_mod_msk;
    ora_mulop inv_mtilde_mod_Bsk[ORA_MAXP];
    uint64_t neg_inv_prod_q_mod_mtilde;
    uint64_t prod_q_mod_Bsk[ORA_MAXP];
    /* decrypt ({t,gamma} base) */
    ora_mulop prod_t_gamma_mod_q[ORA_MAXP];
    ora_mulop neg_inv_q_mod_t_gamma[2];
    ora_mulop inv_gamma_mod_t;
    /* key level: q_special^{-1} mod q_i */
    ora_mulop inv_q_last_mod_q[ORA_MAXP];
    /* plaintext lift / scaling (SEAL context data) */
    uint64_t coeff_div_plain[ORA_MAXP];     /* floor(q/t) mod q_i                      */
    uint64_t q_mod_t;                       /* q mod t                                  */
    uint64_t upper_half_threshold;          /* (t+1)/2                                  */
    uint64_t upper_half_increment[ORA_MAXP];/* q_i - t  (fast plain lift)               */
    int fast_plain_lift;
    uint32_t *batch_index_map;              /* BatchEncoder matrix_reps_index_map       */
};

/* ntt helpers */
int ora_ntt_init(ora_ntt *t, int logn, uint64_t q);
void ora_ntt_free(ora_ntt *t);
void ora_ntt_fwd(const ora_ntt *t, uint64_t *x);
void ora_ntt_inv(const ora_ntt *t, uint64_t *x);
uint64_t ora_invmod(uint64_t a, const ora_mod *m); /* a^{-1} mod m (m prime or a odd & m=2^k) */
void ora_baseconv_init(ora_baseconv *bc, const ora_mod *in, size_t in_n, const ora_mod *out, size_t out_n);
/* in: [in_n][n] (stride n) -> out [out_n][n] */
void ora_baseconv_apply(const ora_baseconv *bc, const uint64_t *in, uint64_t *out, size_t n);

/* PRNG (client side only) */
uint64_t ora_rand64(void);

#endif
