// sunscreen_amd/csrc/kernels.hpp -- host-callable launchers for the kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "devctx.hpp"
#include "rng.hpp"

namespace hipbfv {

// Descriptor tables of the graph executor's table-driven kernels (program.cpp); they live in device memory for one run.
// One output of an n-ary signed sum: out u64[batch][size][K][N] = sum over terms[first .. first + count)
struct NaryOut {
  u64* out;
  u32 first, count, size, pad;
};
// One term: ciphertexts u64[batch][size][K][N]; polynomials beyond `size` count as zero; sign +1 / -1
struct NaryTerm {
  const u64* ptr;
  u32 size;
  int sign;
};
// A plaintext in transform form: u64[K][N] at ptr + item * stride (stride 0: one plaintext for the whole batch)
struct PlainNttRef {
  const u64* ptr;
  u64 stride;
};

// Per-item key selection of one key-switch launch.  The reference passes the keys with every call (sunscreen_runtime/src/run.rs:100-105,
// runtime.rs:310-327), so a server that batches the calls of many clients holds one key set per client.  keys == nullptr: every item
// of the launch uses the launch's one key.  Otherwise position w of the launch's walk handles item order[w].x with the key
// keys[order[w].y]; `order` lists the items of the launch sorted by key, so that neighbours in the walk share key rows in L2.
struct KeyMap {
  const u64* const* keys = nullptr;  // device table of key base pointers (u64[K][2][K+1][N] each)
  const uint2* order = nullptr;      // device, one {item, key index} entry per item of the launch
};

// Which modulus a residue polynomial of a batched buffer belongs to:
// modulus id of polynomial p = mod[(p / div) % period].
struct NttPlan {
  u32 div;
  u32 period;
  unsigned char mod[kMaxMod];
};

hipError_t launch_ntt(const DevCtx* ctx, const MulOp* tw, u32 logn, u64* data, size_t polys, const NttPlan& plan, bool inverse, int scale_mode, hipStream_t s);
hipError_t launch_behz_extend(const DevCtx* ctx, u32 n, u32 K, const u64* in0, u32 sa, const u64* in1, u32 sb, size_t ops, u64* out, hipStream_t s);
hipError_t launch_tensor(const DevCtx* ctx, u32 n, u32 R, const u64* ext, u32 sa, u32 sb, u64* D, size_t ops, hipStream_t s);
hipError_t launch_behz_floor_sk(const DevCtx* ctx, u32 n, u32 K, const u64* D, u64* out, size_t polys, hipStream_t s);
hipError_t launch_ks_decompose(const DevCtx* ctx, u32 n, u32 K, const u64* target, size_t tstride, u64* T, size_t ops, hipStream_t s);
hipError_t launch_ks_mac(const DevCtx* ctx, u32 n, u32 KK, const u64* T, const u64* key, u64* ACC, size_t ops, hipStream_t s, KeyMap km = KeyMap{});
// two-kernel stand-alone transforms for N = 32768 (kernels_split.hip)
hipError_t launch_ntt_split(const DevCtx* ctx, const MulOp* tw, u32 logn, u64* data, size_t polys, const NttPlan& plan, bool inverse,
                            int scale_mode, hipStream_t s);
// split (head / middle / tail) key switch, kernels_split.hip
// res_* (here and in launch_mul_mid): device byte lists of residue indices, read through the scalar unit as 32-bit words -- they must be
// 4-byte aligned and readable up to the next word boundary (the DevCtx members are; the launchers refuse anything else)
// ginv != 0 (launch_ks_head, launch_ks_tail): a rotation's key switch -- the target / the base polynomials are sigma_g(.) for g = ginv^-1 mod 2N,
// read through the automorphism instead of from a rotated copy (base and out2 must not alias then)
hipError_t launch_ks_head(const DevCtx* ctx, const MulOp* twf, u32 logn, int pack, bool mixed, u32 K, const u64* target, size_t tstride, u64* T, size_t ops, hipStream_t s,
                          u32 ginv = 0);
// h: the host copy of *ctx (its residue lists' counts: ks_nd 8-byte FP64 rows, ks_ndp packed FP64 rows, ks_ni integer rows)
hipError_t launch_ks_mid(const DevCtx* ctx, const MulOp* twf, const MulOp* twi, u32 logn, const DevCtx& h, const u64* T, const u64* key, u64* ACC, size_t ops,
                         hipStream_t s, KeyMap km = KeyMap{});
hipError_t launch_ks_tail(const DevCtx* ctx, const MulOp* twi, u32 logn, int pack, bool mixed, const u64* ACC, const u64* base, size_t bstride, u32 base_mask, const u64* extra, u64* out2,
                          size_t ops, hipStream_t s, u32 ginv = 0);
// split BEHZ multiply (2 x 2 -> 3), K <= 4
// Per-member operands of a MERGED multiply launch (graph executor): item i of the launch is item (first + i) % per of member
// (first + i) / per, whose two factors are read where they are (u64[per][2][K][N] each; b == a for a squaring launch) -- no gather
struct MemberHead {
  const u64* a;
  const u64* b;
};
hipError_t launch_mul_head(const DevCtx* ctx, const MulOp* twf, u32 logn, bool aux_f64, int pack, u32 kneed, const u64* a, const u64* b, u64* ext, size_t ops,
                           hipStream_t s, u32 npolys = 4, const MemberHead* members = nullptr, u32 first = 0, u32 per = 0);
hipError_t launch_mul_mid(const DevCtx* ctx, const MulOp* twf, const MulOp* twi, u32 logn, const unsigned char* res_dp, u32 ndp, const unsigned char* res_d, u32 nd,
                          const unsigned char* res_i, u32 ni, const u64* ext, u64* D, size_t ops, hipStream_t s, bool square = false);
hipError_t launch_mul_tail(const DevCtx* ctx, const MulOp* twi, u32 logn, bool aux_f64, int pack, bool conv_grid, u32 kneed, const u64* D, u64* out,
                           size_t ops, hipStream_t s, u32 poly0 = 0, u32 npolys = 3);
hipError_t launch_mulrelin_head(const DevCtx* ctx, const MulOp* twi, const MulOp* twf, u32 logn, int pack_mul, bool conv_grid, int pack_ks, u32 kneed,
                                const u64* D, u64* T, size_t ops, hipStream_t s);
// Per-member epilogue of a MERGED multiply + relinearize launch (the graph executor's members x batch items): item i of the launch is
// item (first + i) % per of member (first + i) / per, and the last kernel writes  mult * product + sign * extra  (mod q, canonical)
// into that member's own buffer -- the Add / Sub chain around a product (examples/chi_sq: 2 x^2, 4 n0 n2 - n1^2) without a pass of
// its own, and program outputs written where the caller wants them.  A device table; all-FP64 fused path only (mulrelin_tail_kernel).
struct MemberTail {
  u64* out;          // u64[per][2][K][N]
  const u64* extra;  // u64[per][2][K][N]; nullptr with sign == 0
  u32 mult;          // 1 ... 4
  int sign;          // -1, 0, +1
};
hipError_t launch_mulrelin_tail(const DevCtx* ctx, const MulOp* twi, u32 logn, int pack_mul, bool conv_grid, int pack_ks, u32 kneed, const u64* D,
                                const u64* ACC, const u64* extra, u64* out2, size_t ops, hipStream_t s, const MemberTail* members = nullptr, u32 first = 0, u32 per = 0);
// the same two kernels for MIXED contexts (integer-policy data / key primes + the FP64 auxiliary base, K <= 4)
hipError_t launch_mulrelin_head_mixed(const DevCtx* ctx, const MulOp* twi, const MulOp* twf, u32 logn, const u64* D, u64* T, size_t ops, hipStream_t s);
hipError_t launch_mulrelin_tail_mixed(const DevCtx* ctx, const MulOp* twi, u32 logn, const u64* D, const u64* ACC, const u64* extra, u64* out2, size_t ops,
                                      hipStream_t s);
hipError_t launch_ks_moddown(const DevCtx* ctx, u32 n, const u64* ACC, const u64* base, size_t bstride, u32 base_mask, const u64* extra, u64* out,
                             size_t ops, hipStream_t s);
hipError_t launch_mod_switch(const DevCtx* ctx, u32 n, const u64* in, u64* out, size_t polys, hipStream_t s);
hipError_t launch_galois(const DevCtx* ctx, u32 n, u32 K, const u64* in, u64* out, size_t polys, u32 ginv, hipStream_t s);
hipError_t launch_eltwise(const DevCtx* ctx, u32 n, const u64* a, const u64* b, u64* out, size_t residue_polys, int mode, hipStream_t s);
hipError_t launch_plain_addsub(const DevCtx* ctx, u32 n, u64* ct, size_t ctstride, const u64* plain, size_t pstride, size_t ops, int sub, hipStream_t s);
hipError_t launch_plain_lift(const DevCtx* ctx, u32 n, const u64* plain, size_t pstride, u64* out, size_t ops, u32* nonzero, hipStream_t s);
hipError_t launch_ct_plain(const DevCtx* ctx, const MulOp* twf, const MulOp* twi, u32 logn, u32 K, bool any_d, bool any_i, const u64* pn, size_t pnstride,
                           const u64* in, u64* out, u32 size, size_t ops, hipStream_t s);
hipError_t launch_dyadic_plain(const DevCtx* ctx, u32 n, u32 K, u64* x, u32 size, const u64* pl, size_t plstride, size_t ops, hipStream_t s);
hipError_t launch_mono_mul(const DevCtx* ctx, u32 n, const u64* in, u64* out, size_t residue_polys, const u64* coeff_rns, u32 e, hipStream_t s);
// graph executor (program.cpp): every Add / Sub / Negate tree of a scheduling round as ONE launch over descriptor tables
hipError_t launch_nary_sum(const DevCtx* ctx, u32 n, u32 K, const NaryOut* outs, const NaryTerm* terms, u32 nouts, u32 max_size, u32 batch, hipStream_t s);
// sum_j ct_j (.) plain[row][j] in the transform domain with the plaintexts behind a pointer table: ctn u64[cols][batch][2][K][N],
// tab [rows][cols], acc u64[rows][batch][2][K][N] (still in transform form)
hipError_t launch_dot_plain_tab(const DevCtx* ctx, u32 n, u32 K, const u64* ctn, u32 cols, const PlainNttRef* tab, u32 rows, u32 batch, u64* acc, hipStream_t s);
// combined handle-level calls (capi.cpp): tables of device pointers / flags live in pinned, device-addressable host memory
hipError_t launch_gather_items(const u64* const* table, u64* stage, size_t words, size_t items, hipStream_t s);
hipError_t launch_copy_words(const u64* src, u64* dst, size_t words, hipStream_t s);  // src may be pinned, device-addressable host memory
hipError_t launch_scatter_items(const u64* stage, u64* const* table, size_t words, size_t items, hipStream_t s);
hipError_t launch_eltwise_items(const DevCtx* ctx, u32 n, u32 K, const u64* const* ta, const u64* const* tb, u64* const* tout, int mode, size_t items,
                                hipStream_t s);
hipError_t launch_transparent_flags_items(const u64* const* table, size_t words_per_ct, size_t skip_words, u32* host_flags, size_t items, hipStream_t s);
hipError_t launch_transparent_flags(const u64* ct, size_t words_per_ct, size_t skip_words, u32* host_flags, size_t items, hipStream_t s);
hipError_t launch_transparent_flag(const u64* ct, size_t words, size_t skip_words, u32* host_flag, hipStream_t s);
hipError_t launch_transparent_watch_nary(const DevCtx* ctx, const NaryOut* outs, u32 nouts, u32 batch, u32* status, hipStream_t s);
hipError_t launch_zero_plain_watch(const u32* nonzero, u32 first_item, u32 item_step, u32 count, u32* status, hipStream_t s);
hipError_t launch_transparent_watch(const u64* ct, size_t words_per_ct, size_t skip_words, u32 first_item, u32* status, size_t ops, hipStream_t s);
hipError_t launch_nonzero_tail(const u64* ct, size_t words_per_ct, size_t skip_words, u32* flags, size_t ops, hipStream_t s);

// kernels_client.hip: BatchEncoder / Decryptor / Encryptor (coefficient-parallel parts)
hipError_t launch_batch_scatter(const DevCtx* ctx, u32 n, const u32* map, const u64* values, u64* plain, size_t ops, int is_signed, u32* bad,
                                hipStream_t s);
hipError_t launch_batch_gather(const DevCtx* ctx, u32 n, const u32* map, const u64* tmp, u64* values, size_t ops, int is_signed, hipStream_t s);
hipError_t launch_dot_secret(const DevCtx* ctx, u32 n, u32 K, const u64* ctn, u32 size, const u64* sk, u64* acc, size_t ops, hipStream_t s);
hipError_t launch_decrypt_round(const DevCtx* ctx, u32 n, const u64* ct, u32 size, const u64* acc, u64* plain, size_t ops, hipStream_t s);
hipError_t launch_encrypt_sample(const DevCtx* ctx, u32 n, const RngSeed& seed, u64 op0, u64* u, u64* e, size_t ops, hipStream_t s);
hipError_t launch_ntt_inv_dyadic(const DevCtx* ctx, const MulOp* tw_inv, u32 logn, const u64* a, const u64* b, u64* c, u32 nmod, u32 nb, u32 bstride,
                                 size_t ops, hipStream_t s);
hipError_t launch_encrypt_finish(const DevCtx* ctx, u32 n, const RngSeed& seed, u64 op0, const u64* c2, const u64* plain, size_t pstride, u64* out, size_t ops,
                                 hipStream_t s);
hipError_t launch_encrypt_dyadic(const DevCtx* ctx, u32 n, u32 KK, const u64* un, const u64* pk, u64* c, size_t ops, hipStream_t s);
hipError_t launch_keygen_ternary(const DevCtx* ctx, u32 n, const RngSeed& seed, u64 stream, u64* s_out, hipStream_t s);
hipError_t launch_keygen_sample(const DevCtx* ctx, u32 n, const RngSeed& seed, u64 stream0, u64* a, u64* e, size_t count, hipStream_t s);
hipError_t launch_keygen_assemble(const DevCtx* ctx, u32 n, u32 KK, const u64* a, const u64* e, const u64* sk, const u64* w, u64* key, size_t count,
                                  hipStream_t s);
hipError_t launch_keygen_square(const DevCtx* ctx, u32 n, u32 KK, const u64* in, u64* out, hipStream_t s);
hipError_t launch_keygen_galois(const DevCtx* ctx, u32 n, u32 KK, const u64* in, u64* out, u32 ginv, hipStream_t s);
hipError_t launch_dot_plain(const DevCtx* ctx, u32 n, u32 K, const u64* ctn, u32 cols, const u64* pntt, u32 rows, u64* acc, hipStream_t s);
hipError_t launch_crt_compose(const DevCtx* ctx, u32 n, u32 KC, const u64* consts, const u64* in, u64* out, u32 polys, hipStream_t s);
hipError_t launch_crt_decompose(const DevCtx* ctx, u32 n, u32 KC, const u64* in, u64* out, u32 polys, hipStream_t s);
hipError_t launch_add_key_level(const DevCtx* ctx, u32 n, u64* c, const u64* e, size_t residue_polys, hipStream_t s);

}  // namespace hipbfv
