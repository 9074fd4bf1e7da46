/*
 * include/hipbfv.h -- C ABI of libhipbfv.so, the MI355X-native BFV ciphertext-arithmetic backend.
 *
 * This is the drop-in boundary for the one hot path of the reference: everything that
 * `seal_fhe::Evaluator` (seal_fhe/src/evaluator.rs:7-280) forwards to Microsoft SEAL's C API
 * through bindgen.  Part 1 re-exports the SEAL C entry points that the `seal_fhe` crate binds for
 * that path, with the same names, argument order, opaque `void*` handles and HRESULT return values,
 * so that the files under `seal_fhe/src/` can link against this library unchanged.
 * Part 2 is the extension the reference API lacks: raw-array import/export for handles and the
 * batched, device-pointer entry points used by the GPU batch executor.
 *
 * Conventions (seal_fhe/src/lib.rs:28-34, error.rs:65-91):
 *   - every function returns a `long` HRESULT: S_OK = 0 on success;
 *   - objects are opaque `void*` created through an out-parameter and destroyed by X_Destroy;
 *   - `pool` arguments are accepted and ignored (the crate always passes NULL);
 *   - in-place calls alias `destination` with an operand (evaluator_base.rs:107-113,184-196);
 *   - all handle-level calls are synchronous and may be issued concurrently from several host
 *     threads on one evaluator handle (sunscreen_runtime/src/run.rs:415-469).
 * Ciphertext data layout: u64[size][K][N], K = data-level primes, N = poly degree
 * (seal_fhe/src/plaintext_ciphertext.rs:303-314).  Key-switching key: u64[K][2][K+1][N], NTT form.
 */
#ifndef HIPBFV_H
#define HIPBFV_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIPBFV_S_OK 0L
#define HIPBFV_E_POINTER ((long)0x80004003L)
#define HIPBFV_E_INVALIDARG ((long)0x80070057L)
#define HIPBFV_E_OUTOFMEMORY ((long)0x8007000EL)
#define HIPBFV_E_UNEXPECTED ((long)0x8000FFFFL)
#define HIPBFV_COR_E_IO ((long)0x80131620L)
#define HIPBFV_COR_E_INVALIDOPERATION ((long)0x80131509L)

/* ===================================================================================== */
/* Part 1: SEAL C API subset bound by seal_fhe for the evaluator path                      */
/* ===================================================================================== */

/* ---- Modulus (seal_fhe/src/modulus.rs:225-262) ---- */
long Modulus_Create1(uint64_t value, void **small_modulus);
long Modulus_Create2(void *copy, void **small_modulus);
long Modulus_Destroy(void *thisptr);
long Modulus_Value(void *thisptr, uint64_t *result);

/* ---- CoeffModulus helpers (seal_fhe/src/modulus.rs:149-205) ---- */
long CoeffModulus_MaxBitCount(uint64_t poly_modulus_degree, int sec_level, int *bit_count);
long CoeffModulus_BFVDefault(uint64_t poly_modulus_degree, int sec_level, uint64_t *length, void **coeffs);
long CoeffModulus_Create1(uint64_t poly_modulus_degree, uint64_t length, int *bit_sizes, void **coeffs);
/* PlainModulus::batching is CoeffModulus_Create1 with one entry (seal_fhe/src/modulus.rs:100-116) */

/* ---- EncryptionParameters (seal_fhe/src/encryption_parameters.rs:100-330); scheme 1 = BFV ---- */
long EncParams_Create1(uint8_t scheme, void **enc_params);
long EncParams_Destroy(void *thisptr);
long EncParams_SetPolyModulusDegree(void *thisptr, uint64_t degree);
long EncParams_GetPolyModulusDegree(void *thisptr, uint64_t *degree);
long EncParams_SetCoeffModulus(void *thisptr, uint64_t length, void **coeffs);
long EncParams_GetCoeffModulus(void *thisptr, uint64_t *length, void **coeffs);
long EncParams_SetPlainModulus1(void *thisptr, void *modulus);
long EncParams_SetPlainModulus2(void *thisptr, uint64_t plain_modulus);
long EncParams_GetPlainModulus(void *thisptr, void **plain_modulus);
long EncParams_GetScheme(void *thisptr, uint8_t *scheme);

/* ---- SEALContext (seal_fhe/src/context.rs:63-115); sec_level: 0 none, 128, 192, 256 ---- */
long SEALContext_Create(void *encryption_params, bool expand_mod_chain, int sec_level, void **context);
long SEALContext_Destroy(void *thisptr);

/* ---- Plaintext (seal_fhe/src/plaintext_ciphertext.rs:36-300) ---- */
long Plaintext_Create1(void *pool, void **plaintext);
long Plaintext_Create4(char *hex_poly, void *pool, void **plaintext); /* "7FFx^3 + 1x^1 + 3", plaintext_ciphertext.rs:180-217 */
long Plaintext_Create5(void *copy, void **plaintext);
long Plaintext_Destroy(void *thisptr);
long Plaintext_CoeffCount(void *thisptr, uint64_t *coeff_count);
long Plaintext_CoeffAt(void *thisptr, uint64_t index, uint64_t *coeff);
long Plaintext_SetCoeffAt(void *thisptr, uint64_t index, uint64_t value);
long Plaintext_Resize(void *thisptr, uint64_t coeff_count);
long Plaintext_IsNTTForm(void *thisptr, bool *is_ntt_form);
/* SEAL 4.0 wire format; compr_mode 0 = none, 2 = zstd (seal_fhe/src/plaintext_ciphertext.rs:100-160) */
long Plaintext_SaveSize(void *thisptr, uint8_t compr_mode, int64_t *result);
long Plaintext_Save(void *thisptr, uint8_t *outptr, uint64_t size, uint8_t compr_mode, int64_t *out_bytes);
long Plaintext_Load(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes);

/* ---- Ciphertext (seal_fhe/src/plaintext_ciphertext.rs:326-504) ---- */
long Ciphertext_Create1(void *pool, void **cipher);
long Ciphertext_Create2(void *copy, void **cipher);
long Ciphertext_Destroy(void *thisptr);
long Ciphertext_Size(void *thisptr, uint64_t *size);
long Ciphertext_CoeffModulusSize(void *thisptr, uint64_t *coeff_modulus_size);
long Ciphertext_PolyModulusDegree(void *thisptr, uint64_t *poly_modulus_degree);
long Ciphertext_GetDataAt1(void *thisptr, uint64_t index, uint64_t *data);
long Ciphertext_GetDataAt2(void *thisptr, uint64_t poly_index, uint64_t coeff_index, uint64_t *data);
long Ciphertext_IsNTTForm(void *thisptr, bool *is_ntt_form);
/* SEAL 4.0 wire format (seal_fhe/src/plaintext_ciphertext.rs:451-497) */
long Ciphertext_SaveSize(void *thisptr, uint8_t compr_mode, int64_t *result);
long Ciphertext_Save(void *thisptr, uint8_t *outptr, uint64_t size, uint8_t compr_mode, int64_t *out_bytes);
long Ciphertext_Load(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes);

/* ---- KSwitchKeys: RelinKeys and GaloisKeys (seal_fhe/src/key_generator.rs:467-729) ---- */
long KSwitchKeys_Create1(void **kswitch_keys);
long KSwitchKeys_Create2(void *copy, void **kswitch_keys);
long KSwitchKeys_Destroy(void *thisptr);
/* SEAL 4.0 wire format (seal_fhe/src/key_generator.rs:493-573, 649-729) */
long KSwitchKeys_SaveSize(void *thisptr, uint8_t compr_mode, int64_t *result);
long KSwitchKeys_Save(void *thisptr, uint8_t *outptr, uint64_t size, uint8_t compr_mode, int64_t *out_bytes);
long KSwitchKeys_Load(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes);

/* ---- SecretKey / PublicKey (seal_fhe/src/key_generator.rs:200-430): handles + SEAL 4.0 wire format.
 * Key data is key-level NTT form:
 * SecretKey u64[K+1][N], PublicKey u64[2][K+1][N]. ---- */
long SecretKey_Create1(void **key);
long SecretKey_Create2(void *copy, void **key);
long SecretKey_Destroy(void *thisptr);
long SecretKey_SaveSize(void *thisptr, uint8_t compr_mode, int64_t *result);
long SecretKey_Save(void *thisptr, uint8_t *outptr, uint64_t size, uint8_t compr_mode, int64_t *out_bytes);
long SecretKey_Load(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes);
long PublicKey_Create1(void **key);
long PublicKey_Create2(void *copy, void **key);
long PublicKey_Destroy(void *thisptr);
long PublicKey_SaveSize(void *thisptr, uint8_t compr_mode, int64_t *result);
long PublicKey_Save(void *thisptr, uint8_t *outptr, uint64_t size, uint8_t compr_mode, int64_t *out_bytes);
long PublicKey_Load(void *thisptr, void *context, uint8_t *inptr, uint64_t size, int64_t *in_bytes);

/* ---- KeyGenerator (seal_fhe/src/key_generator.rs:20-200): keys are sampled and assembled on the device (Philox4x32-10;
 * ternary secret, uniform a, rounded Gaussian errors; key-switching keys in SEAL's layout u64[K][2][K+1][N]).  Like
 * SEAL's, the keys are not reproducible across implementations; they interoperate through the wire format.
 * save_seed (seed-compressed serialisation) is accepted and ignored: handles always hold expanded keys. ---- */
long KeyGenerator_Create1(void *context, void **key_generator);
long KeyGenerator_Create2(void *context, void *secret_key, void **key_generator);
long KeyGenerator_Destroy(void *thisptr);
long KeyGenerator_SecretKey(void *thisptr, void **secret_key);
long KeyGenerator_CreatePublicKey(void *thisptr, bool save_seed, void **public_key);
long KeyGenerator_CreateRelinKeys(void *thisptr, bool save_seed, void **relin_keys);
long KeyGenerator_CreateGaloisKeysFromElts(void *thisptr, uint64_t count, uint32_t *galois_elts, bool save_seed, void **galois_keys);
long KeyGenerator_CreateGaloisKeysAll(void *thisptr, bool save_seed, void **galois_keys);
long KeyGenerator_CreateGaloisKeysFromSteps(void *thisptr, uint64_t count, int *steps, bool save_seed, void **galois_keys); /* SEAL keygenerator.h; not bound by seal_fhe */

/* ---- BatchEncoder (seal_fhe/src/encoder.rs:50-215): slot vectors <-> plaintexts, transforms over Z_t on the device ---- */
long BatchEncoder_Create(void *context, void **encoder);
long BatchEncoder_Destroy(void *thisptr);
long BatchEncoder_Encode1(void *thisptr, uint64_t count, uint64_t *values, void *destination);
long BatchEncoder_Encode2(void *thisptr, uint64_t count, int64_t *values, void *destination);
long BatchEncoder_Decode1(void *thisptr, void *plain, uint64_t *count, uint64_t *destination, void *pool);
long BatchEncoder_Decode2(void *thisptr, void *plain, uint64_t *count, int64_t *destination, void *pool);
long BatchEncoder_GetSlotCount(void *thisptr, uint64_t *slot_count);

/* ---- Decryptor (seal_fhe/src/encryptor_decryptor.rs:596-690): <ct, (1, s, s^2)> then SEAL's
 * decrypt_scale_and_round in the base {t, gamma}; bit-identical plaintexts ---- */
long Decryptor_Create(void *context, void *secret_key, void **decryptor);
long Decryptor_Destroy(void *thisptr);
long Decryptor_Decrypt(void *thisptr, void *encrypted, void *destination);
long Decryptor_InvariantNoiseBudget(void *thisptr, void *encrypted, int *invariant_noise_budget);
long Decryptor_InvariantNoise(void *thisptr, void *encrypted, double *invariant_noise); /* fork: encryptor_decryptor.rs:660-683 */

/* ---- Encryptor (seal_fhe/src/encryptor_decryptor.rs:140-600).  The randomness is the library's own (Philox4x32-10;
 * ternary u, rounded Gaussian sigma 3.2 clipped at 19, uniform a): ciphertexts are valid SEAL ciphertexts but, like
 * SEAL's, not reproducible across implementations.  public_key or secret_key may be NULL (with_public_key /
 * with_secret_key / with_public_and_secret_key).  The *ReturnComponents* entry points are the Sunscreen fork's
 * (encryptor_decryptor.rs:268-590): they also hand back the sampled u (1 polynomial), e (2; 1 in symmetric mode) as
 * PolynomialArrays over the data primes and the scaling remainder r, such that over the data primes, exactly,
 *   public key, disable_special_modulus: c0 = floor(q/t) m + r + pk0 u + e0,  c1 = pk1 u + e1
 *   secret key:                          c0 = floor(q/t) m + r - (c1 s + e)
 * (logproof/src/bfv_statement.rs:159-160).  The *SetSeed variants take the fork's [u64; 8] seed, folded into the
 * Philox key: equal seeds give equal ciphertexts (not SEAL's bits: its PRNG is not restated).  save_seed is ignored. ---- */
long Encryptor_Create(void *context, void *public_key, void *secret_key, void **encryptor);
long Encryptor_Destroy(void *thisptr);
long Encryptor_Encrypt(void *thisptr, void *plaintext, void *destination, void *pool);
long Encryptor_EncryptReturnComponents(void *thisptr, void *plaintext, bool disable_special_modulus, void *destination,
                                       void *u_destination, void *e_destination, void *r_destination, void *pool);
long Encryptor_EncryptReturnComponentsSetSeed(void *thisptr, void *plaintext, bool disable_special_modulus, void *destination,
                                              void *u_destination, void *e_destination, void *r_destination, void *seed, void *pool);
long Encryptor_EncryptSymmetric(void *thisptr, void *plaintext, bool save_seed, void *destination, void *pool);
long Encryptor_EncryptSymmetricReturnComponents(void *thisptr, void *plaintext, void *destination, void *e_destination,
                                                void *r_destination, void *pool);
long Encryptor_EncryptSymmetricReturnComponentsSetSeed(void *thisptr, void *plaintext, void *destination, void *e_destination,
                                                       void *r_destination, void *seed, void *pool);

/* ---- PolynomialArray, the Sunscreen fork's export type (seal_fhe/src/data_structures.rs:17-304): `PolySize`
 * coefficient-form polynomials over the first `CoeffModulusSize` primes, exported either as RNS u64[poly][rns][coeff]
 * or, after ToMultiprecision, as u64[poly][coeff][limb] (CRT-composed value in [0, q), little-endian limbs) -- the layout
 * logproof/src/bfv_statement.rs:624-660 consumes.  Keys are converted out of NTT form and restricted to the data primes.
 * The conversions run on the device.  Drop keeps residues 0..k-2 of every polynomial. ---- */
long PolynomialArray_Create(void *pool, void **poly_array);
long PolynomialArray_CreateFromCiphertext(void *pool, void *context, void *ciphertext, void **poly_array);
long PolynomialArray_CreateFromPublicKey(void *pool, void *context, void *public_key, void **poly_array);
long PolynomialArray_CreateFromSecretKey(void *pool, void *context, void *secret_key, void **poly_array);
long PolynomialArray_Copy(void *copy, void **poly_array);
long PolynomialArray_Destroy(void *thisptr);
long PolynomialArray_IsReserved(void *thisptr, bool *is_reserved);
long PolynomialArray_IsRns(void *thisptr, bool *is_rns);
long PolynomialArray_ToRns(void *thisptr);
long PolynomialArray_ToMultiprecision(void *thisptr);
long PolynomialArray_PolySize(void *thisptr, uint64_t *size);
long PolynomialArray_PolyModulusDegree(void *thisptr, uint64_t *size);
long PolynomialArray_CoeffModulusSize(void *thisptr, uint64_t *size);
long PolynomialArray_ExportSize(void *thisptr, uint64_t *size);
long PolynomialArray_PerformExport(void *thisptr, uint64_t *data);
long PolynomialArray_Drop(void *thisptr, void **poly_array);

/* ---- Evaluator (seal_fhe/src/evaluator_base.rs:55-407, bfv_evaluator.rs:12-248) ---- */
long Evaluator_Create(void *seal_context, void **evaluator);
long Evaluator_Destroy(void *thisptr);
long Evaluator_Negate(void *thisptr, void *encrypted, void *destination);
long Evaluator_Add(void *thisptr, void *encrypted1, void *encrypted2, void *destination);
long Evaluator_AddMany(void *thisptr, uint64_t count, void **encrypteds, void *destination);
long Evaluator_Sub(void *thisptr, void *encrypted1, void *encrypted2, void *destination);
long Evaluator_Multiply(void *thisptr, void *encrypted1, void *encrypted2, void *destination, void *pool);
long Evaluator_MultiplyMany(void *thisptr, uint64_t count, void **encrypteds, void *relin_keys, void *destination, void *pool);
/* Square = Multiply(x, x) bit for bit; when the two operands of a multiply are ONE ciphertext (here, Evaluator_Multiply with
 * encrypted1 == encrypted2, hipbfv_batch_multiply* with a == b, a program node whose operands are one node) the kernels extend
 * and transform it once: two forward transforms per residue instead of four. */
long Evaluator_Square(void *thisptr, void *encrypted, void *destination, void *pool);
long Evaluator_Relinearize(void *thisptr, void *encrypted, void *relin_keys, void *destination, void *pool);
long Evaluator_Exponentiate(void *thisptr, void *encrypted, uint64_t exponent, void *relin_keys, void *destination, void *pool);
long Evaluator_AddPlain(void *thisptr, void *encrypted, void *plain, void *destination);
long Evaluator_SubPlain(void *thisptr, void *encrypted, void *plain, void *destination);
long Evaluator_MultiplyPlain(void *thisptr, void *encrypted, void *plain, void *destination, void *pool);
long Evaluator_RotateRows(void *thisptr, void *encrypted, int steps, void *galois_keys, void *destination, void *pool);
long Evaluator_RotateColumns(void *thisptr, void *encrypted, void *galois_keys, void *destination, void *pool);
/* mod_switch_to_next (seal_fhe/src/evaluator.rs:84-157): BFV ciphertexts are divided-and-rounded by the last data prime
 * and move to the next level of the modulus-switching chain; every Evaluator_*, Decryptor_* and Ciphertext_Save/Load
 * call accepts ciphertexts of any level (levels are created on first use; SEALContext_Create(expand_mod_chain = false)
 * disables them).  Plaintexts of this path are never in NTT form, so the plaintext overload always fails as SEAL's does. */
long Evaluator_ModSwitchToNext1(void *thisptr, void *encrypted, void *destination, void *pool);
long Evaluator_ModSwitchToNext2(void *thisptr, void *plain, void *destination);

/* ===================================================================================== */
/* Part 2: hipbfv extensions                                                               */
/* ===================================================================================== */

/* Library / device */
long hipbfv_version(uint32_t *major, uint32_t *minor);
long hipbfv_last_error(char *buffer, uint64_t capacity);       /* thread-local message of the last failure */
long hipbfv_build_flags(char *buffer, uint64_t capacity);      /* compiler flags (-D set included) this library was built with */
long hipbfv_set_device(int device);                            /* HIP device used by contexts created afterwards */
/* SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT switch (seal_fhe `transparent-ciphertexts` feature): default on */
long hipbfv_set_throw_on_transparent(bool enabled);

/* Context shortcuts: build a context straight from raw parameters */
long hipbfv_Context_Create(uint64_t poly_modulus_degree, const uint64_t *coeff_modulus, uint64_t coeff_count,
                           uint64_t plain_modulus, void **context);
long hipbfv_Context_Info(void *context, uint64_t *poly_modulus_degree, uint64_t *data_primes, uint64_t *key_primes,
                         uint64_t *plain_modulus);
long hipbfv_Context_GetPrime(void *context, uint64_t index, uint64_t *value); /* key-level prime `index` */
/* the context of the next level of the modulus-switching chain (one data prime fewer, same special prime): an owned
 * handle for the batched API (Evaluator_Create on it, hipbfv_batch_* at that level) */
long hipbfv_Context_NextLevel(void *context, void **next);
/* The BEHZ auxiliary base Bsk = B u {m_sk} this context multiplies in (internal to Evaluator_Multiply; SEAL's
 * RNSTool keeps its own privately, native/src/seal/util/rns.h).  *fp64_base bit 0 = the library chose its own
 * FP64-pipe primes (same size bound as SEAL's rule, bit-identical products) instead of SEAL's 61-bit primes;
 * bit 1 / bit 2 = the split multiply / key-switch pipelines store their intermediates 48-bit packed; bit 3 = the
 * multiply forms its q -> Bsk base-conversion sums exactly and reduces each once; bit 4 = mixed base: the data primes are too
 * wide for the FP64 pipe (the 3 x 54-bit set) but the auxiliary primes are the library's own FP64-pipe ones; bit 5 = the multiply packs
 * its intermediates PER ROW: only the rows whose prime is below 2^48 (the SEAL default set of n = 16384: 13 of 18 rows); bit 6 = the key
 * switch packs its digit rows per KEY PRIME (bit 2 is set too; n = 16384: the rows of 3 of the 9 key primes) (diagnostics).
 * primes may be NULL; otherwise capacity >= *count words, B first, m_sk last. */
long hipbfv_Context_AuxBase(void *context, uint64_t *count, uint64_t *primes, uint64_t capacity, int *fp64_base);

/* Raw-array import/export for handles (host memory): the no-serialisation alternative to X_Load / X_Save */
long hipbfv_Ciphertext_Assign(void *cipher, void *context, uint64_t size, const uint64_t *host_data);
long hipbfv_Ciphertext_Export(void *cipher, uint64_t *host_data, uint64_t capacity_words);
long hipbfv_Ciphertext_DevicePtr(void *cipher, uint64_t **device_ptr);
long hipbfv_KSwitchKeys_AssignRelin(void *keys, void *context, const uint64_t *host_data);
long hipbfv_KSwitchKeys_AssignGalois(void *keys, void *context, uint32_t galois_elt, const uint64_t *host_data);
long hipbfv_KSwitchKeys_DevicePtr(void *keys, uint64_t index, uint64_t **device_ptr); /* index 0 = relin, (elt-1)/2 = galois */
long hipbfv_KSwitchKeys_Read(void *keys, uint64_t index, uint64_t *host_out);          /* u64[K][2][K+1][N] */
long hipbfv_KSwitchKeys_Has(void *keys, uint64_t index, bool *present);
long hipbfv_SecretKey_Read(void *key, uint64_t *host_out);                              /* u64[K+1][N] */
long hipbfv_PublicKey_Read(void *key, uint64_t *host_out);                              /* u64[2][K+1][N] */
long hipbfv_SecretKey_Assign(void *key, void *context, const uint64_t *host_data);  /* u64[K+1][N], NTT form */
long hipbfv_PublicKey_Assign(void *key, void *context, const uint64_t *host_data);  /* u64[2][K+1][N], NTT form */
long hipbfv_Encryptor_SetSeed(void *encryptor, uint64_t seed);                      /* reproducible runs (tests) */
long hipbfv_KeyGenerator_SetSeed(void *key_generator, uint64_t seed);               /* keys created afterwards */
long hipbfv_KeyGenerator_CreateSeeded(void *context, uint64_t seed, void **key_generator); /* reproducible secret */

/* Host-only helpers of the SEAL 4.0 wire format (no device access; used by the CPU tests against the
 * reference's binary fixtures): parms_id = BLAKE2b-256([scheme=1, n, primes..., t]); decode/encode one object. */
long hipbfv_wire_parms_id(uint64_t poly_modulus_degree, const uint64_t *primes, uint64_t count, uint64_t plain_modulus,
                          uint8_t *out32);
long hipbfv_wire_decode_ciphertext(const uint8_t *in, uint64_t in_size, uint8_t *parms_id32, bool *is_ntt, uint64_t *size,
                                   uint64_t *poly_modulus_degree, uint64_t *coeff_modulus_size, uint64_t *data,
                                   uint64_t capacity_words, int64_t *in_bytes);
long hipbfv_wire_encode_ciphertext(const uint8_t *parms_id32, bool is_ntt, uint64_t size, uint64_t poly_modulus_degree,
                                   uint64_t coeff_modulus_size, const uint64_t *data, uint8_t compr_mode, uint8_t *out,
                                   uint64_t capacity, int64_t *out_bytes);
long hipbfv_wire_decode_plaintext(const uint8_t *in, uint64_t in_size, uint8_t *parms_id32, uint64_t *coeff_count,
                                  uint64_t *coeffs, uint64_t capacity_words, int64_t *in_bytes);

/* Batched entry points: device pointers, `count` independent ciphertexts u64[count][size][K][N],
 * enqueued asynchronously on `stream` (a hipStream_t; NULL = default stream). */
long hipbfv_batch_multiply(void *evaluator, const uint64_t *a, uint64_t size_a, const uint64_t *b, uint64_t size_b,
                           uint64_t *out, uint64_t count, void *stream);
long hipbfv_batch_relinearize(void *evaluator, const uint64_t *ct3, void *relin_keys, uint64_t *out2, uint64_t count,
                              void *stream);
long hipbfv_batch_multiply_relin(void *evaluator, const uint64_t *a, const uint64_t *b, void *relin_keys,
                                 uint64_t *out2, uint64_t count, void *stream);
long hipbfv_batch_apply_galois(void *evaluator, const uint64_t *ct2, uint32_t galois_elt, void *galois_keys,
                               uint64_t *out2, uint64_t count, void *stream);
long hipbfv_batch_rotate_rows(void *evaluator, const uint64_t *ct2, int steps, void *galois_keys, uint64_t *out2,
                              uint64_t count, void *stream);
long hipbfv_batch_rotate_columns(void *evaluator, const uint64_t *ct2, void *galois_keys, uint64_t *out2,
                                 uint64_t count, void *stream);
/* Per-key batches (multi-tenant serving).  The reference hands the keys over with every call
 * (sunscreen_runtime/src/run.rs:100-105: `relin_keys: &Option<&RelinearizationKeys>`, `galois_keys: &Option<&GaloisKeys>`;
 * runtime.rs:310-327), so a server that batches the calls of many clients holds one key set per client.  These are the
 * three key-switching operations above with `num_key_sets` key handles (RelinearizationKeys resp. GaloisKeys objects of the
 * evaluator's context) and, per item, the set it uses: key_index is a HOST array of `count` entries < num_key_sets, read
 * before the call returns.  Item i gives the bits of the single-key call with key_sets[key_index[i]].  Items need not be
 * grouped by key: the key-switch kernel walks them in key order, so items of one client share that client's key rows in
 * one XCD's L2; with one key set per item the key (16 K (K+1) N bytes) is part of every item's compulsory traffic.
 * rotate_rows_keys follows SEAL's rotate_internal: the direct key when EVERY set holds it, the NAF chain otherwise. */
long hipbfv_batch_relinearize_keys(void *evaluator, const uint64_t *ct3, void *const *relin_key_sets, uint64_t num_key_sets,
                                   const uint32_t *key_index, uint64_t *out2, uint64_t count, void *stream);
long hipbfv_batch_multiply_relin_keys(void *evaluator, const uint64_t *a, const uint64_t *b, void *const *relin_key_sets,
                                      uint64_t num_key_sets, const uint32_t *key_index, uint64_t *out2, uint64_t count,
                                      void *stream);
long hipbfv_batch_apply_galois_keys(void *evaluator, const uint64_t *ct2, uint32_t galois_elt, void *const *galois_key_sets,
                                    uint64_t num_key_sets, const uint32_t *key_index, uint64_t *out2, uint64_t count,
                                    void *stream);
long hipbfv_batch_rotate_rows_keys(void *evaluator, const uint64_t *ct2, int steps, void *const *galois_key_sets,
                                   uint64_t num_key_sets, const uint32_t *key_index, uint64_t *out2, uint64_t count,
                                   void *stream);
long hipbfv_batch_rotate_columns_keys(void *evaluator, const uint64_t *ct2, void *const *galois_key_sets,
                                      uint64_t num_key_sets, const uint32_t *key_index, uint64_t *out2, uint64_t count,
                                      void *stream);
long hipbfv_batch_add(void *evaluator, const uint64_t *a, const uint64_t *b, uint64_t *out, uint64_t size,
                      uint64_t count, void *stream);
long hipbfv_batch_sub(void *evaluator, const uint64_t *a, const uint64_t *b, uint64_t *out, uint64_t size,
                      uint64_t count, void *stream);
long hipbfv_batch_negate(void *evaluator, const uint64_t *a, uint64_t *out, uint64_t size, uint64_t count, void *stream);
/* plain: device u64[count][N] (plain_stride = N) or one shared plaintext (plain_stride = 0) */
long hipbfv_batch_add_plain(void *evaluator, const uint64_t *ct, uint64_t size, const uint64_t *plain,
                            uint64_t plain_stride, uint64_t *out, uint64_t count, void *stream);
long hipbfv_batch_sub_plain(void *evaluator, const uint64_t *ct, uint64_t size, const uint64_t *plain,
                            uint64_t plain_stride, uint64_t *out, uint64_t count, void *stream);
long hipbfv_batch_multiply_plain(void *evaluator, const uint64_t *ct, uint64_t size, const uint64_t *plain,
                                 uint64_t plain_stride, uint64_t *out, uint64_t count, void *stream);
/* forward / inverse negacyclic NTT of u64[polys][N]; polynomial p uses key-level prime (p % nprimes) */
long hipbfv_batch_ntt(void *evaluator, uint64_t *data, uint64_t polys, uint64_t nprimes, bool inverse, void *stream);
/* ct u64[count][size][K][N] -> out u64[count][size][K-1][N] (the layout of hipbfv_Context_NextLevel's context) */
long hipbfv_batch_mod_switch(void *evaluator, const uint64_t *ct, uint64_t size, uint64_t *out, uint64_t count, void *stream);
/* The steps either side of the evaluator, batched on device buffers:
 * encode / decode: values u64[count][N] (int64 when is_signed) <-> plaintexts u64[count][N];
 * decrypt: ct u64[count][size][K][N] -> plaintexts u64[count][N] (zero padded);
 * encrypt: plaintexts u64[count][N] (plain_stride = N) or one shared (0) -> ct u64[count][2][K][N]; item i draws its
 * randomness from the ChaCha20 blocks at position first_op + i under the seed (distinct (seed, op) pairs give independent
 * randomness; the generator is described in sunscreen_amd/csrc/rng.hpp).  hipbfv_batch_encrypt_seeded takes the 512-bit
 * seed SEAL's prng_seed_type carries (64 bytes; NULL = 512 fresh bits from getrandom(2), failing with E_UNEXPECTED if
 * the OS has none) and is the production entry point.  hipbfv_batch_encrypt with its 64-bit `seed` is TEST ONLY
 * (reproducible batches for the parity tests: 2^64 candidates can be searched), like hipbfv_Encryptor_SetSeed,
 * hipbfv_KeyGenerator_SetSeed and hipbfv_KeyGenerator_CreateSeeded.
 * encode synchronises the stream (it reports out-of-range values); the others are asynchronous. */
long hipbfv_batch_encode(void *evaluator, const uint64_t *values, uint64_t *plain, uint64_t count, int is_signed, void *stream);
long hipbfv_batch_decode(void *evaluator, const uint64_t *plain, uint64_t *values, uint64_t count, int is_signed, void *stream);
long hipbfv_batch_decrypt(void *evaluator, const uint64_t *ct, uint32_t size, void *secret_key, uint64_t *plain, uint64_t count, void *stream);
long hipbfv_batch_encrypt(void *evaluator, const uint64_t *plain, uint64_t plain_stride, void *public_key, uint64_t seed, uint64_t first_op,
                          uint64_t *ct, uint64_t count, void *stream);
long hipbfv_batch_encrypt_seeded(void *evaluator, const uint64_t *plain, uint64_t plain_stride, void *public_key, const uint8_t *seed64,
                                 uint64_t first_op, uint64_t *ct, uint64_t count, void *stream);
/* Plaintext-matrix x ciphertext-vector products in the transform domain -- the first loop nest of examples/pir
 * (examples/pir/src/main.rs:16-45: col[i] = sum_j database[i][j] * col_query[j]).  plain_to_ntt transforms plaintexts
 * the way Evaluator_MultiplyPlain does internally (a static database is transformed once); ct_to_ntt transforms every
 * polynomial of the ciphertexts; dot_plain_ntt produces, for size-2 ciphertexts, out[row] = sum_j MultiplyPlain(ct_j,
 * plain[row][j]) in coefficient form, bit-identical to the reference's sequence of multiply_plain and add.
 * ctn: u64[cols][2][K][N], pntt: u64[rows][cols][K][N], out: u64[rows][2][K][N].
 * An ALL-ZERO plaintext has no transformed form: SEAL refuses every product with it (transparent result), and the consumers
 * of `pntt` cannot see it any more, so plain_to_ntt records the index of the first all-zero plaintext in the evaluator's
 * status word -- hipbfv_batch_status returns COR_E_INVALIDOPERATION for it, as for any transparent batch result. */
long hipbfv_batch_plain_to_ntt(void *evaluator, const uint64_t *plain, uint64_t plain_stride, uint64_t *pntt, uint64_t count, void *stream);
long hipbfv_batch_ct_to_ntt(void *evaluator, const uint64_t *ct, uint64_t size, uint64_t *ctn, uint64_t count, void *stream);
long hipbfv_batch_dot_plain_ntt(void *evaluator, const uint64_t *ctn, uint64_t cols, const uint64_t *pntt, uint64_t rows, uint64_t *out,
                                void *stream);
long hipbfv_set_chunk_ops(void *evaluator, uint64_t chunk_ops);

/* Batch executor for compiled FHE program graphs (replaces sunscreen_runtime/src/run.rs:100-357).
 * Node kinds follow sunscreen_fhe_program::Operation (operation.rs:12-94) in this order:
 * 0 ShiftLeft, 1 ShiftRight, 2 SwapRows, 3 Relinearize, 4 Multiply, 5 MultiplyPlaintext, 6 Add,
 * 7 AddPlaintext, 8 Negate, 9 Sub, 10 SubPlaintext, 11 InputCiphertext(arg), 12 InputPlaintext(arg),
 * 13 Literal::U64(arg), 14 OutputCiphertext (15 Literal::Plaintext: hipbfv_Program_AddPlaintextLiteral / JSON only).
 * Edge kinds: 0 Left, 1 Right, 2 Unary.
 * LoadJson accepts the serde JSON form of `FheProgram` (petgraph StableGraph).
 * Run executes the graph over `batch` independent input sets: input i is a device pointer to
 * u64[batch][2][K][N] (kind 0), to plaintexts u64[batch][N] / one shared u64[N] (kind 1, stride N / 0), or to plaintexts
 * already lifted and transformed, u64[batch][K][N] / one shared u64[K][N] (kind 2, stride K*N / 0: the output of
 * hipbfv_batch_plain_to_ntt -- static data such as examples/pir's database is transformed once, not per query; only
 * MultiplyPlaintext nodes may consume it, and its zero check is the producer's: hipbfv_batch_plain_to_ntt + hipbfv_batch_status);
 * one output buffer u64[batch][2][K][N] per OutputCiphertext node, in node order.
 * Execution follows the reference's `traverse` (sunscreen_runtime/src/run.rs:372-472: every node whose operands are
 * complete runs at once): ready nodes of one kind are one batched launch, Add / Sub / Negate trees are n-ary sums, sums of
 * ciphertext-plaintext products stay in the transform domain (program_plan.cpp).  Bits are those of the node-by-node
 * evaluation; HIPBFV_PROGRAM_SERIAL=1 selects that executor (one node at a time, kinds 0 / 1 only).
 *
 * Transparent results (SEAL built with SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT, seal_fhe/build.rs:46-66: `runtime.run` fails
 * on `a * 0`, sunscreen/tests/features.rs:8-34).  The handle-level Evaluator_* functions check their one result before
 * they return.  The batched paths check on the device: every operation records, in a status word, the smallest batch
 * index whose result is transparent.  Program_Run owns such a word per run, reads it once at exit (one stream
 * synchronisation per run) and returns COR_E_INVALIDOPERATION -- hipbfv_last_error names AN input set whose result is
 * transparent (merged launches number their ciphertexts member-major, so it need not be the lowest such set).  The
 * hipbfv_batch_* operations stay asynchronous on `stream` and record into their evaluator's word; hipbfv_batch_status
 * synchronises `stream`, reads and resets it: COR_E_INVALIDOPERATION + the first transparent item (of any operation since
 * the previous call), S_OK + ~0 otherwise.  hipbfv_set_batch_transparent_check(evaluator, false) switches the recording
 * off for one evaluator; hipbfv_set_throw_on_transparent(false) for the whole library (the crate's
 * `transparent-ciphertexts` feature). */
long hipbfv_Program_Create(void **program);
long hipbfv_Program_Destroy(void *program);
long hipbfv_Program_AddNode(void *program, uint32_t op, uint64_t arg, uint32_t *node_id);
/* Literal::Plaintext(bytes) node: `bytes` is what the compiler stores in the graph -- bincode of
 * InnerPlaintext::Seal([WithContext{Params, SEAL-serialised Plaintext}]) (sunscreen_fhe_program/src/literal.rs:8-18,
 * sunscreen/src/fhe/mod.rs:370-376, decoded by the reference at sunscreen_runtime/src/run.rs:312-328).  The
 * parameters inside are checked against the evaluator's context when the program runs. */
long hipbfv_Program_AddPlaintextLiteral(void *program, const uint8_t *bytes, uint64_t length, uint32_t *node_id);
long hipbfv_Program_AddEdge(void *program, uint32_t src, uint32_t dst, uint32_t kind);
long hipbfv_Program_LoadJson(void *program, const char *json, uint64_t length);
long hipbfv_batch_status(void *evaluator, uint64_t *first_transparent_item, void *stream);
long hipbfv_set_batch_transparent_check(void *evaluator, bool enabled);
long hipbfv_Program_NumOutputs(void *program, uint64_t *count);
/* Diagnostic: one multiply + relinearize of ONE ciphertext pair (device pointers u64[2][K][N], a relinearisation key
 * u64[K][2][K+1][N], result u64[2][K][N]) timed from the host, launched kernel by kernel and replayed from a captured hipGraph
 * (microseconds per repetition, one stream synchronisation each). */
long hipbfv_debug_graph_probe(void *context, const uint64_t *a, const uint64_t *b, const uint64_t *relin_key, uint64_t *out,
                              uint64_t iterations, double *us_direct, double *us_graph);
/* Diagnostic, host only (no device): the FP64 range plans of one prime at degree 2^log_n -- out6 = {FP64 policy possible,
 * forward reduce mask, inverse reduce mask, split pipelines possible, split forward mask, split inverse mask}; bit p of a mask =
 * every value is reduced mod q at the start of pass p (bit 8 of the split inverse mask: at the start of the tail stages). */
long hipbfv_debug_f64_plan(uint64_t prime, uint32_t log_n, uint32_t *out6);
/* Diagnostic, host only (no device): the auxiliary base a context with these parameters gets -- the same values
 * hipbfv_Context_AuxBase reports for a live context (B, then m_sk last), `flags` likewise.  tests/test_behz_base_bound_cpu.py
 * replays the BEHZ floor and Shenoy-Kumaresan steps in exact integers over the base returned here. */
long hipbfv_debug_aux_base(uint64_t poly_modulus_degree, const uint64_t *coeff_primes, uint64_t prime_count, uint64_t plain_modulus,
                           uint64_t *count, uint64_t *primes, uint64_t capacity, int *flags);
/* The schedule Run follows, one line per step ("mul_relin members=3 square", "sum members=2 terms=6", "plain_matrix members=256
 * columns=256", ...): `*needed` = bytes including the terminator; `buffer` may be NULL to ask for the size. */
long hipbfv_Program_Describe(void *program, char *buffer, uint64_t capacity, uint64_t *needed);
/* Run: one stream synchronisation per 32768 input sets (the status word is read once per such chunk; batches up to 32768 --
 * every BASELINE configuration -- synchronise once, at exit).  A batch above 32768 runs as consecutive chunks over offset
 * pointers: a failure or a transparent result in chunk k returns its error with the outputs of chunks 0 .. k-1 already written. */
long hipbfv_Program_Run(void *program, void *evaluator, uint64_t batch, uint64_t num_inputs, const uint32_t *input_kinds,
                        const uint64_t *const *input_ptrs, const uint64_t *input_strides, void *relin_keys,
                        void *galois_keys, uint64_t num_outputs, uint64_t *const *outputs, void *stream);

/* Run with one key set per client: input set i of the batch uses relin_keys[key_index[i]] / galois_keys[key_index[i]]
 * (HOST arrays: num_key_sets handles each -- an entry may be NULL when the program needs no such key -- and `batch`
 * indices).  Same bits per input set as hipbfv_Program_Run with that set's keys (the reference's call, run.rs:100-105). */
long hipbfv_Program_RunKeys(void *program, void *evaluator, uint64_t batch, uint64_t num_inputs, const uint32_t *input_kinds,
                            const uint64_t *const *input_ptrs, const uint64_t *input_strides, uint64_t num_key_sets,
                            void *const *relin_keys, void *const *galois_keys, const uint32_t *key_index,
                            uint64_t num_outputs, uint64_t *const *outputs, void *stream);

/* Per-kernel timing (HIP events recorded on the launch stream, around every kernel launch):
 * total milliseconds, number of launches and work units (residue polynomials for the NTT kernels,
 * polynomials or operations for the others) since the last reset. */
long hipbfv_profile_enable(void *evaluator, bool enabled);
long hipbfv_profile_reset(void *evaluator);
long hipbfv_profile_kernel_count(uint32_t *count);
long hipbfv_profile_read(void *evaluator, uint32_t kernel_id, char *name, uint64_t name_capacity, double *total_ms,
                         uint64_t *launches, uint64_t *units);

#ifdef __cplusplus
}
#endif
#endif /* HIPBFV_H */
