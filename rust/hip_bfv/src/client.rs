//! The client-side half of the `seal_fhe` surface -- parameters, keys, encoder, encryptor, decryptor -- over the same C ABI
//! (`include/hipbfv.h` Part 1 exports every entry point `seal_fhe` binds for them: `modulus.rs`, `encryption_parameters.rs`,
//! `context.rs`, `key_generator.rs`, `encoder.rs`, `encryptor_decryptor.rs`).  With it `sunscreen_runtime/src/runtime.rs:20-23`
//! (`BfvEncryptionParametersBuilder`, `Context`, `KeyGenerator`, `Encryptor`, `Decryptor`, `Modulus`) resolves against this
//! crate as it does against `seal_fhe`.  Names, argument order and error behaviour are the reference's; the bodies are one C
//! call each.  Key generation, encryption and decryption run on the device.
use std::ffi::c_void;
use std::marker::PhantomData;
use std::ptr::null_mut;

use crate::{bindgen, check, Ciphertext, Context, Error, GaloisKeys, Plaintext, RelinearizationKeys, Result};

/// `seal_fhe::SecurityLevel` (`modulus.rs:40-60`): the HomomorphicEncryption.org levels SEAL enforces at context creation.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
#[repr(i32)]
pub enum SecurityLevel {
    TC128 = 128,
    TC192 = 192,
    TC256 = 256,
}

/// SEAL's three compression modes of `X_Save` (`serialization.rs`): the wire format is SEAL 4.0's.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
#[repr(u8)]
pub enum CompressionType {
    None = 0,
    ZLib = 1,
    ZStd = 2,
}

/// `seal_fhe::ToBytes` / `FromBytes` (`lib.rs:75-100`).
pub trait ToBytes {
    fn as_bytes(&self) -> Result<Vec<u8>>;
}
pub trait FromBytes: Sized {
    fn from_bytes(context: &Context, bytes: &[u8]) -> Result<Self>;
}

/// An owned handle with SEAL-format serialisation: `$save_size` / `$save` / `$load` are the three C entry points.
macro_rules! serialisable {
    ($name:ident, $create:expr, $save_size:ident, $save:ident, $load:ident) => {
        impl ToBytes for $name {
            fn as_bytes(&self) -> Result<Vec<u8>> {
                let mode = CompressionType::ZStd as u8;
                let mut size: i64 = 0;
                check(unsafe { bindgen::$save_size(self.handle, mode, &mut size) })?;
                let mut out = vec![0u8; size as usize];
                let mut written: i64 = 0;
                check(unsafe { bindgen::$save(self.handle, out.as_mut_ptr(), out.len() as u64, mode, &mut written) })?;
                out.truncate(written as usize);
                Ok(out)
            }
        }
        impl FromBytes for $name {
            fn from_bytes(context: &Context, bytes: &[u8]) -> Result<Self> {
                let obj: $name = $create()?;
                let mut read: i64 = 0;
                check(unsafe { bindgen::$load(obj.handle, context.handle, bytes.as_ptr() as *mut u8, bytes.len() as u64, &mut read) })?;
                Ok(obj)
            }
        }
        impl PartialEq for $name {
            /// equality of the serialised bytes, as in the reference (`plaintext_ciphertext.rs:445-449`)
            fn eq(&self, other: &Self) -> bool {
                matches!((self.as_bytes(), other.as_bytes()), (Ok(a), Ok(b)) if a == b)
            }
        }
    };
}

macro_rules! plain_handle {
    ($name:ident, $destroy:ident) => {
        pub struct $name {
            pub(crate) handle: *mut c_void,
        }
        unsafe impl Sync for $name {}
        unsafe impl Send for $name {}
        impl $name {
            pub fn get_handle(&self) -> *mut c_void {
                self.handle
            }
        }
        impl Drop for $name {
            fn drop(&mut self) {
                check(unsafe { bindgen::$destroy(self.handle) }).expect("destroying a libhipbfv object failed");
            }
        }
    };
}

// ---------------------------------------------------------------------------------------------------------------- moduli
plain_handle!(Modulus, Modulus_Destroy);

impl Modulus {
    pub fn new(value: u64) -> Result<Self> {
        let mut handle = null_mut();
        check(unsafe { bindgen::Modulus_Create1(value, &mut handle) })?;
        Ok(Self { handle })
    }
    pub fn value(&self) -> u64 {
        let mut v = 0u64;
        check(unsafe { bindgen::Modulus_Value(self.handle, &mut v) }).expect("Modulus_Value");
        v
    }
    /// Adopt a handle the library allocated for the caller (the arrays `CoeffModulus_*` fill).
    pub(crate) fn adopt(handle: *mut c_void) -> Self {
        Self { handle }
    }
}
impl Clone for Modulus {
    fn clone(&self) -> Self {
        let mut handle = null_mut();
        check(unsafe { bindgen::Modulus_Create2(self.handle, &mut handle) }).expect("Modulus_Create2");
        Self { handle }
    }
}
impl PartialEq for Modulus {
    fn eq(&self, other: &Self) -> bool {
        self.value() == other.value()
    }
}
impl std::fmt::Debug for Modulus {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "Modulus({})", self.value())
    }
}

/// `seal_fhe::CoefficientModulus` (`modulus.rs:160-250`).
pub struct CoefficientModulus;
impl CoefficientModulus {
    /// NTT-friendly primes of the given bit sizes: SEAL hands them out smallest-first within a size class
    /// (pinned by the reference's known answers, `modulus.rs:279-313`).
    pub fn create(degree: u64, bit_sizes: &[i32]) -> Result<Vec<Modulus>> {
        let mut sizes = bit_sizes.to_vec();
        let mut raw: Vec<*mut c_void> = vec![null_mut(); sizes.len()];
        check(unsafe { bindgen::CoeffModulus_Create1(degree, sizes.len() as u64, sizes.as_mut_ptr(), raw.as_mut_ptr()) })?;
        Ok(raw.into_iter().map(Modulus::adopt).collect())
    }
    /// SEAL's default coefficient modulus for a degree and security level.
    pub fn bfv_default(degree: u64, security_level: SecurityLevel) -> Result<Vec<Modulus>> {
        let mut len = 0u64;
        check(unsafe { bindgen::CoeffModulus_BFVDefault(degree, security_level as i32, &mut len, null_mut()) })?;
        let mut raw: Vec<*mut c_void> = vec![null_mut(); len as usize];
        check(unsafe { bindgen::CoeffModulus_BFVDefault(degree, security_level as i32, &mut len, raw.as_mut_ptr()) })?;
        Ok(raw.into_iter().map(Modulus::adopt).collect())
    }
    pub fn max_bit_count(degree: u64, security_level: SecurityLevel) -> u32 {
        let mut bits: i32 = 0;
        check(unsafe { bindgen::CoeffModulus_MaxBitCount(degree, security_level as i32, &mut bits) }).expect("CoeffModulus_MaxBitCount");
        bits as u32
    }
}

/// `seal_fhe::PlainModulus` (`modulus.rs:252-275`).
pub struct PlainModulus;
impl PlainModulus {
    /// A prime of `bit_size` bits congruent to 1 mod 2 * degree (batching); the largest one, as SEAL picks it.
    pub fn batching(degree: u64, bit_size: u32) -> Result<Modulus> {
        let mut v = CoefficientModulus::create(degree, &[bit_size as i32])?;
        v.pop().ok_or_else(|| Error::Unexpected("no batching prime of that size".into()))
    }
    pub fn raw(val: u64) -> Result<Modulus> {
        Modulus::new(val)
    }
}

// ------------------------------------------------------------------------------------------------------------ parameters
plain_handle!(EncryptionParameters, EncParams_Destroy);

/// `seal_fhe::SchemeType` (`encryption_parameters.rs:20-45`); this library implements BFV.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
#[repr(u8)]
pub enum SchemeType {
    None = 0,
    Bfv = 1,
    Ckks = 2,
}

impl EncryptionParameters {
    pub fn get_poly_modulus_degree(&self) -> u64 {
        let mut d = 0u64;
        check(unsafe { bindgen::EncParams_GetPolyModulusDegree(self.handle, &mut d) }).expect("EncParams_GetPolyModulusDegree");
        d
    }
    pub fn get_scheme(&self) -> SchemeType {
        let mut s = 0u8;
        check(unsafe { bindgen::EncParams_GetScheme(self.handle, &mut s) }).expect("EncParams_GetScheme");
        if s == 1 { SchemeType::Bfv } else if s == 2 { SchemeType::Ckks } else { SchemeType::None }
    }
    /// A copy of the plain modulus (the getter's handle is borrowed from the parameters: `encryption_parameters.rs:136-143`).
    pub fn get_plain_modulus(&self) -> Modulus {
        let mut borrowed = null_mut();
        check(unsafe { bindgen::EncParams_GetPlainModulus(self.handle, &mut borrowed) }).expect("EncParams_GetPlainModulus");
        let view = std::mem::ManuallyDrop::new(Modulus::adopt(borrowed));
        (*view).clone()
    }
    pub fn get_coefficient_modulus(&self) -> Vec<Modulus> {
        let mut len = 0u64;
        check(unsafe { bindgen::EncParams_GetCoeffModulus(self.handle, &mut len, null_mut()) }).expect("EncParams_GetCoeffModulus");
        let mut raw: Vec<*mut c_void> = vec![null_mut(); len as usize];
        check(unsafe { bindgen::EncParams_GetCoeffModulus(self.handle, &mut len, raw.as_mut_ptr()) }).expect("EncParams_GetCoeffModulus");
        raw.into_iter().map(Modulus::adopt).collect()
    }
}

/// `seal_fhe::BfvEncryptionParametersBuilder` (`encryption_parameters.rs:190-300`).
#[derive(Default)]
pub struct BfvEncryptionParametersBuilder {
    degree: Option<u64>,
    coefficient: Option<Vec<Modulus>>,
    plain: Option<Modulus>,
}
impl BfvEncryptionParametersBuilder {
    pub fn new() -> Self {
        Self::default()
    }
    pub fn set_poly_modulus_degree(mut self, degree: u64) -> Self {
        self.degree = Some(degree);
        self
    }
    pub fn set_coefficient_modulus(mut self, modulus: Vec<Modulus>) -> Self {
        self.coefficient = Some(modulus);
        self
    }
    pub fn set_plain_modulus(mut self, modulus: Modulus) -> Self {
        self.plain = Some(modulus);
        self
    }
    pub fn set_plain_modulus_u64(mut self, modulus: u64) -> Self {
        self.plain = Modulus::new(modulus).ok();
        self
    }
    /// Fails, as the reference does, when a component was never set.
    pub fn build(self) -> Result<EncryptionParameters> {
        let degree = self.degree.ok_or_else(|| Error::InvalidArgument("polynomial modulus degree not set".into()))?;
        let coefficient = self.coefficient.ok_or_else(|| Error::InvalidArgument("coefficient modulus not set".into()))?;
        let plain = self.plain.ok_or_else(|| Error::InvalidArgument("plain modulus not set".into()))?;
        let mut handle = null_mut();
        check(unsafe { bindgen::EncParams_Create1(SchemeType::Bfv as u8, &mut handle) })?;
        let params = EncryptionParameters { handle };
        check(unsafe { bindgen::EncParams_SetPolyModulusDegree(params.handle, degree) })?;
        let mut handles: Vec<*mut c_void> = coefficient.iter().map(|m| m.get_handle()).collect();
        check(unsafe { bindgen::EncParams_SetCoeffModulus(params.handle, handles.len() as u64, handles.as_mut_ptr()) })?;
        check(unsafe { bindgen::EncParams_SetPlainModulus1(params.handle, plain.get_handle()) })?;
        Ok(params)
    }
}

impl Context {
    /// `seal_fhe::Context::new` (`context.rs:63-80`): tables and keys-level constants are built on the device.
    pub fn new(params: &EncryptionParameters, expand_mod_chain: bool, security_level: SecurityLevel) -> Result<Self> {
        let mut handle = null_mut();
        check(unsafe { bindgen::SEALContext_Create(params.get_handle(), expand_mod_chain, security_level as i32, &mut handle) })?;
        Ok(Self { handle })
    }
    /// No security check (`context.rs:92-100`): SEAL's `sec_level_type::none`.
    pub fn new_insecure(params: &EncryptionParameters, expand_mod_chain: bool) -> Result<Self> {
        let mut handle = null_mut();
        check(unsafe { bindgen::SEALContext_Create(params.get_handle(), expand_mod_chain, 0, &mut handle) })?;
        Ok(Self { handle })
    }
}

// ------------------------------------------------------------------------------------------------------------------ keys
plain_handle!(PublicKey, PublicKey_Destroy);
plain_handle!(SecretKey, SecretKey_Destroy);
plain_handle!(KeyGenerator, KeyGenerator_Destroy);

impl PublicKey {
    pub fn new() -> Result<Self> {
        let mut handle = null_mut();
        check(unsafe { bindgen::PublicKey_Create1(&mut handle) })?;
        Ok(Self { handle })
    }
}
impl Clone for PublicKey {
    fn clone(&self) -> Self {
        let mut handle = null_mut();
        check(unsafe { bindgen::PublicKey_Create2(self.handle, &mut handle) }).expect("PublicKey_Create2");
        Self { handle }
    }
}
impl SecretKey {
    pub fn new() -> Result<Self> {
        let mut handle = null_mut();
        check(unsafe { bindgen::SecretKey_Create1(&mut handle) })?;
        Ok(Self { handle })
    }
}
impl Clone for SecretKey {
    fn clone(&self) -> Self {
        let mut handle = null_mut();
        check(unsafe { bindgen::SecretKey_Create2(self.handle, &mut handle) }).expect("SecretKey_Create2");
        Self { handle }
    }
}
impl std::fmt::Debug for SecretKey {
    /// never prints key material (`key_generator.rs:428-434`)
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.write_str("SecretKey(<redacted>)")
    }
}
serialisable!(PublicKey, PublicKey::new, PublicKey_SaveSize, PublicKey_Save, PublicKey_Load);
serialisable!(SecretKey, SecretKey::new, SecretKey_SaveSize, SecretKey_Save, SecretKey_Load);
serialisable!(Ciphertext, Ciphertext::new, Ciphertext_SaveSize, Ciphertext_Save, Ciphertext_Load);
serialisable!(Plaintext, Plaintext::new, Plaintext_SaveSize, Plaintext_Save, Plaintext_Load);
serialisable!(RelinearizationKeys, RelinearizationKeys::empty, KSwitchKeys_SaveSize, KSwitchKeys_Save, KSwitchKeys_Load);
serialisable!(GaloisKeys, GaloisKeys::new, KSwitchKeys_SaveSize, KSwitchKeys_Save, KSwitchKeys_Load);

impl KeyGenerator {
    /// A fresh ternary secret key, sampled on the device from the OS-seeded ChaCha20 stream.
    pub fn new(ctx: &Context) -> Result<Self> {
        let mut handle = null_mut();
        check(unsafe { bindgen::KeyGenerator_Create1(ctx.handle, &mut handle) })?;
        Ok(Self { handle })
    }
    pub fn new_from_secret_key(ctx: &Context, secret_key: &SecretKey) -> Result<Self> {
        let mut handle = null_mut();
        check(unsafe { bindgen::KeyGenerator_Create2(ctx.handle, secret_key.handle, &mut handle) })?;
        Ok(Self { handle })
    }
    pub fn secret_key(&self) -> SecretKey {
        let mut handle = null_mut();
        check(unsafe { bindgen::KeyGenerator_SecretKey(self.handle, &mut handle) }).expect("KeyGenerator_SecretKey");
        SecretKey { handle }
    }
    pub fn create_public_key(&self) -> PublicKey {
        let mut handle = null_mut();
        check(unsafe { bindgen::KeyGenerator_CreatePublicKey(self.handle, false, &mut handle) }).expect("KeyGenerator_CreatePublicKey");
        PublicKey { handle }
    }
    /// Fails when the parameters have a single coefficient prime (no special prime to switch through), as SEAL does.
    pub fn create_relinearization_keys(&self) -> Result<RelinearizationKeys> {
        let mut handle = null_mut();
        check(unsafe { bindgen::KeyGenerator_CreateRelinKeys(self.handle, false, &mut handle) })?;
        Ok(RelinearizationKeys { handle })
    }
    /// Keys for every power-of-two row rotation and the column swap (SEAL's default set).
    pub fn create_galois_keys(&self) -> Result<GaloisKeys> {
        let mut handle = null_mut();
        check(unsafe { bindgen::KeyGenerator_CreateGaloisKeysAll(self.handle, false, &mut handle) })?;
        Ok(GaloisKeys { handle })
    }
    /// Keys for exactly these signed row-rotation steps (0 = the column swap): SEAL's `create_galois_keys(steps)`, which
    /// `seal_fhe` does not bind -- a 14-key set instead of 27 for examples/dot_prod at n = 16384.
    pub fn create_galois_keys_from_steps(&self, steps: &[i32]) -> Result<GaloisKeys> {
        let mut s = steps.to_vec();
        let mut handle = null_mut();
        check(unsafe { bindgen::KeyGenerator_CreateGaloisKeysFromSteps(self.handle, s.len() as u64, s.as_mut_ptr(), false, &mut handle) })?;
        Ok(GaloisKeys { handle })
    }
}

// --------------------------------------------------------------------------------------------------------------- encoder
plain_handle!(BFVEncoder, BatchEncoder_Destroy);

impl BFVEncoder {
    /// `seal_fhe::BFVEncoder` (`encoder.rs:50-215`): SEAL's BatchEncoder -- n slots as a 2 x n/2 matrix of values mod t.
    pub fn new(ctx: &Context) -> Result<Self> {
        let mut handle = null_mut();
        check(unsafe { bindgen::BatchEncoder_Create(ctx.handle, &mut handle) })?;
        Ok(Self { handle })
    }
    pub fn get_slot_count(&self) -> usize {
        let mut n = 0u64;
        check(unsafe { bindgen::BatchEncoder_GetSlotCount(self.handle, &mut n) }).expect("BatchEncoder_GetSlotCount");
        n as usize
    }
    pub fn encode_unsigned(&self, data: &[u64]) -> Result<Plaintext> {
        let plain = Plaintext::new()?;
        let mut v = data.to_vec();
        check(unsafe { bindgen::BatchEncoder_Encode1(self.handle, v.len() as u64, v.as_mut_ptr(), plain.handle) })?;
        Ok(plain)
    }
    pub fn encode_signed(&self, data: &[i64]) -> Result<Plaintext> {
        let plain = Plaintext::new()?;
        let mut v = data.to_vec();
        check(unsafe { bindgen::BatchEncoder_Encode2(self.handle, v.len() as u64, v.as_mut_ptr(), plain.handle) })?;
        Ok(plain)
    }
    pub fn decode_unsigned(&self, plaintext: &Plaintext) -> Result<Vec<u64>> {
        let mut out = vec![0u64; self.get_slot_count()];
        let mut count = 0u64;
        check(unsafe { bindgen::BatchEncoder_Decode1(self.handle, plaintext.handle, &mut count, out.as_mut_ptr(), null_mut()) })?;
        out.truncate(count as usize);
        Ok(out)
    }
    pub fn decode_signed(&self, plaintext: &Plaintext) -> Result<Vec<i64>> {
        let mut out = vec![0i64; self.get_slot_count()];
        let mut count = 0u64;
        check(unsafe { bindgen::BatchEncoder_Decode2(self.handle, plaintext.handle, &mut count, out.as_mut_ptr(), null_mut()) })?;
        out.truncate(count as usize);
        Ok(out)
    }
}

// ------------------------------------------------------------------------------------------------- encryptor / decryptor
/// Marker types of `seal_fhe::Encryptor<T>` (`encryptor_decryptor.rs:82-135`): which keys the encryptor holds decides which
/// methods exist.
pub struct Sym;
pub struct Asym;
pub struct SymAsym;
pub mod enc_marker {
    pub trait Sym {}
    pub trait Asym {}
    impl Sym for super::Sym {}
    impl Sym for super::SymAsym {}
    impl Asym for super::Asym {}
    impl Asym for super::SymAsym {}
}

pub struct Encryptor<T = ()> {
    handle: *mut c_void,
    _keys: PhantomData<T>,
}
unsafe impl<T> Sync for Encryptor<T> {}
unsafe impl<T> Send for Encryptor<T> {}
pub type SymmetricEncryptor = Encryptor<Sym>;
pub type AsymmetricEncryptor = Encryptor<Asym>;
pub type SymAsymEncryptor = Encryptor<SymAsym>;

impl Encryptor {
    pub fn with_public_and_secret_key(ctx: &Context, public_key: &PublicKey, secret_key: &SecretKey) -> Result<SymAsymEncryptor> {
        let mut handle = null_mut();
        check(unsafe { bindgen::Encryptor_Create(ctx.handle, public_key.handle, secret_key.handle, &mut handle) })?;
        Ok(Encryptor { handle, _keys: PhantomData })
    }
    pub fn with_public_key(ctx: &Context, public_key: &PublicKey) -> Result<AsymmetricEncryptor> {
        let mut handle = null_mut();
        check(unsafe { bindgen::Encryptor_Create(ctx.handle, public_key.handle, null_mut(), &mut handle) })?;
        Ok(Encryptor { handle, _keys: PhantomData })
    }
    pub fn with_secret_key(ctx: &Context, secret_key: &SecretKey) -> Result<SymmetricEncryptor> {
        let mut handle = null_mut();
        check(unsafe { bindgen::Encryptor_Create(ctx.handle, null_mut(), secret_key.handle, &mut handle) })?;
        Ok(Encryptor { handle, _keys: PhantomData })
    }
}
impl<T> Encryptor<T> {
    pub fn get_handle(&self) -> *mut c_void {
        self.handle
    }
}
impl<T: enc_marker::Asym> Encryptor<T> {
    /// Public-key encryption on the device: u ternary, e Gaussian (sigma 3.2, clipped), divide-and-round by the special prime.
    pub fn encrypt(&self, plaintext: &Plaintext) -> Result<Ciphertext> {
        let ct = Ciphertext::new()?;
        check(unsafe { bindgen::Encryptor_Encrypt(self.handle, plaintext.handle, ct.handle, null_mut()) })?;
        Ok(ct)
    }
}
impl<T: enc_marker::Sym> Encryptor<T> {
    pub fn encrypt_symmetric(&self, plaintext: &Plaintext) -> Result<Ciphertext> {
        let ct = Ciphertext::new()?;
        check(unsafe { bindgen::Encryptor_EncryptSymmetric(self.handle, plaintext.handle, false, ct.handle, null_mut()) })?;
        Ok(ct)
    }
}
impl<T> Drop for Encryptor<T> {
    fn drop(&mut self) {
        check(unsafe { bindgen::Encryptor_Destroy(self.handle) }).expect("Encryptor_Destroy");
    }
}

plain_handle!(Decryptor, Decryptor_Destroy);
impl Decryptor {
    pub fn new(ctx: &Context, secret_key: &SecretKey) -> Result<Self> {
        let mut handle = null_mut();
        check(unsafe { bindgen::Decryptor_Create(ctx.handle, secret_key.handle, &mut handle) })?;
        Ok(Self { handle })
    }
    pub fn decrypt(&self, ciphertext: &Ciphertext) -> Result<Plaintext> {
        let plain = Plaintext::new()?;
        check(unsafe { bindgen::Decryptor_Decrypt(self.handle, ciphertext.handle, plain.handle) })?;
        Ok(plain)
    }
    /// Bits of noise budget left (0 = decryption no longer reliable): `assumptions.rs:138-194` relies on relinearisation
    /// leaving it unchanged.
    pub fn invariant_noise_budget(&self, ciphertext: &Ciphertext) -> Result<u32> {
        let mut bits: i32 = 0;
        check(unsafe { bindgen::Decryptor_InvariantNoiseBudget(self.handle, ciphertext.handle, &mut bits) })?;
        Ok(bits.max(0) as u32)
    }
    pub fn invariant_noise(&self, ciphertext: &Ciphertext) -> Result<f64> {
        let mut noise = 0f64;
        check(unsafe { bindgen::Decryptor_InvariantNoise(self.handle, ciphertext.handle, &mut noise) })?;
        Ok(noise)
    }
}
