//! `hip_bfv`: BFV ciphertext arithmetic on AMD MI355X behind the surface of `seal_fhe`.
//!
//! The types and the [`Evaluator`] trait carry the names `sunscreen_runtime` already uses
//! (`seal_fhe/src/evaluator.rs:7-280`; the runtime needs `E: Evaluator + Sync + Send`,
//! `sunscreen_runtime/src/run.rs:100`), and `client.rs` carries the parameter / key / encoder / encryptor / decryptor types
//! `runtime.rs:20-23` imports, so switching a build from `seal_fhe` to this crate is a `use` change for the BFV path (the
//! fork-only `PolynomialArray` / `*Components` API that logproof consumes is exported by the C ABI but not wrapped here).
//! Underneath is the C ABI of `include/hipbfv.h`: every handle is an opaque pointer owned by exactly one Rust value
//! whose `Drop` calls `X_Destroy`; `Clone` is a deep copy; errors are SEAL's HRESULTs (`seal_fhe/src/lib.rs:28-34`).
//!
//! Two levels:
//! * handle level ([`BFVEvaluator`] and friends): one ciphertext per call, synchronous, thread-safe -- the drop-in;
//! * batch level ([`batch::BatchEvaluator`], [`batch::Program`]): device-resident `u64[count][size][K][N]` batches,
//!   asynchronous on a HIP stream -- what replaces the per-node dispatch of `run_program_unchecked`.
#![allow(non_upper_case_globals, non_camel_case_types, non_snake_case, dead_code)]

use std::os::raw::c_long;

#[allow(clippy::all)]
pub(crate) mod bindgen {
    include!(concat!(env!("OUT_DIR"), "/bindings.rs"));
}

pub mod batch;
mod client;
mod evaluator;
mod handles;
mod raw;

pub use client::{
    enc_marker, Asym, AsymmetricEncryptor, BFVEncoder, BfvEncryptionParametersBuilder, CoefficientModulus, CompressionType, Decryptor,
    EncryptionParameters, Encryptor, FromBytes, KeyGenerator, Modulus, PlainModulus, PublicKey, SchemeType, SecretKey, SecurityLevel, Sym,
    SymAsym, SymAsymEncryptor, SymmetricEncryptor, ToBytes,
};
pub use evaluator::{BFVEvaluator, Evaluator};
pub use handles::{Ciphertext, Context, GaloisKeys, Plaintext, RelinearizationKeys};

/// SEAL's HRESULT values as the C layer returns them (`seal_fhe/src/lib.rs:28-34`).
pub const E_OK: c_long = 0;
pub const E_POINTER: c_long = 0x8000_4003u32 as i32 as c_long;
pub const E_INVALIDARG: c_long = 0x8007_0057u32 as i32 as c_long;
pub const E_OUTOFMEMORY: c_long = 0x8007_000Eu32 as i32 as c_long;
pub const E_UNEXPECTED: c_long = 0x8000_FFFFu32 as i32 as c_long;
pub const COR_E_IO: c_long = 0x8013_1620u32 as i32 as c_long;
pub const COR_E_INVALIDOPERATION: c_long = 0x8013_1509u32 as i32 as c_long;

/// Same variants as `seal_fhe::Error` (`seal_fhe/src/error.rs:10-78`) so that `?` in the runtime keeps compiling; the
/// library's thread-local message (`hipbfv_last_error`) rides along.
#[derive(Debug, Clone, PartialEq, thiserror::Error)]
pub enum Error {
    #[error("The argument is not valid: {0}")]
    InvalidArgument(String),
    #[error("Invalid pointer")]
    InvalidPointer,
    #[error("Out of memory")]
    OutOfMemory,
    #[error("Unexpected: {0}")]
    Unexpected(String),
    #[error("Internal error {0:#x}: {1}")]
    InternalError(c_long, String),
    #[error("Unknown {0:#x}")]
    Unknown(c_long),
}

pub type Result<T> = std::result::Result<T, Error>;

fn last_error() -> String {
    let mut buf = [0u8; 256];
    unsafe { bindgen::hipbfv_last_error(buf.as_mut_ptr() as *mut _, buf.len() as u64) };
    let end = buf.iter().position(|&b| b == 0).unwrap_or(buf.len());
    String::from_utf8_lossy(&buf[..end]).into_owned()
}

/// HRESULT -> `Result` (the crate-wide analogue of `convert_seal_error`, `seal_fhe/src/error.rs:82-91`).
pub(crate) fn check(hr: c_long) -> Result<()> {
    match hr {
        E_OK => Ok(()),
        E_POINTER => Err(Error::InvalidPointer),
        E_INVALIDARG => Err(Error::InvalidArgument(last_error())),
        E_OUTOFMEMORY => Err(Error::OutOfMemory),
        E_UNEXPECTED => Err(Error::Unexpected(last_error())),
        COR_E_IO | COR_E_INVALIDOPERATION => Err(Error::InternalError(hr, last_error())),
        other => Err(Error::Unknown(other)),
    }
}

/// Pick the HIP device (one process per GPU; must precede the first `Context`).
pub fn set_device(device: i32) -> Result<()> {
    check(unsafe { bindgen::hipbfv_set_device(device) })
}

#[cfg(feature = "transparent-ciphertexts")]
#[ctor::ctor]
fn allow_transparent() {
    unsafe { bindgen::hipbfv_set_throw_on_transparent(false) };
}
