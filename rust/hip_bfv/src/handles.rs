//! Owned handles: one Rust value per C object, `Drop` = `X_Destroy`, `Clone` = `X_Create2` deep copy
//! (the ownership rules of `seal_fhe/src/plaintext_ciphertext.rs:36-52,326-342,499-504`).
use std::ffi::c_void;
use std::ptr::null_mut;

use crate::{bindgen, check, Result};

macro_rules! owned_handle {
    ($name:ident, $destroy:ident) => {
        pub struct $name {
            pub(crate) handle: *mut c_void,
        }
        // the library serialises what needs serialising; every call may come from any thread (seal_fhe/src/lib.rs:7-9)
        unsafe impl Sync for $name {}
        unsafe impl Send for $name {}
        impl $name {
            pub fn get_handle(&self) -> *mut c_void {
                self.handle
            }
        }
        impl Drop for $name {
            fn drop(&mut self) {
                // seal_fhe panics when a destructor fails (evaluator_base.rs:62-67); so does this crate
                check(unsafe { bindgen::$destroy(self.handle) }).expect("destroying a libhipbfv object failed");
            }
        }
    };
}

owned_handle!(Context, SEALContext_Destroy);
owned_handle!(Ciphertext, Ciphertext_Destroy);
owned_handle!(Plaintext, Plaintext_Destroy);
owned_handle!(RelinearizationKeys, KSwitchKeys_Destroy);
owned_handle!(GaloisKeys, KSwitchKeys_Destroy);

impl Context {
    /// Straight from numbers (the library's extension; `SEALContext_Create` over `EncParams_*` works as in seal_fhe).
    pub fn from_raw(poly_modulus_degree: u64, coeff_modulus: &[u64], plain_modulus: u64) -> Result<Self> {
        let mut handle = null_mut();
        check(unsafe {
            bindgen::hipbfv_Context_Create(poly_modulus_degree, coeff_modulus.as_ptr(), coeff_modulus.len() as u64, plain_modulus, &mut handle)
        })?;
        Ok(Self { handle })
    }

    /// (poly_modulus_degree N, data primes K, key primes K + 1, plain modulus): the sizes every raw array is checked against.
    pub fn info(&self) -> Result<(u64, u64, u64, u64)> {
        let (mut n, mut k, mut kk, mut t) = (0u64, 0u64, 0u64, 0u64);
        check(unsafe { bindgen::hipbfv_Context_Info(self.handle, &mut n, &mut k, &mut kk, &mut t) })?;
        Ok((n, k, kk, t))
    }
}

impl Ciphertext {
    pub fn new() -> Result<Self> {
        let mut handle = null_mut();
        check(unsafe { bindgen::Ciphertext_Create1(null_mut(), &mut handle) })?;
        Ok(Self { handle })
    }
    pub fn num_polynomials(&self) -> u64 {
        let mut v = 0;
        check(unsafe { bindgen::Ciphertext_Size(self.handle, &mut v) }).expect("Ciphertext_Size");
        v
    }
    pub fn coeff_modulus_size(&self) -> u64 {
        let mut v = 0;
        check(unsafe { bindgen::Ciphertext_CoeffModulusSize(self.handle, &mut v) }).expect("Ciphertext_CoeffModulusSize");
        v
    }
}

impl Clone for Ciphertext {
    fn clone(&self) -> Self {
        let mut handle = null_mut();
        check(unsafe { bindgen::Ciphertext_Create2(self.handle, &mut handle) }).expect("Ciphertext_Create2");
        Self { handle }
    }
}

impl Plaintext {
    pub fn new() -> Result<Self> {
        let mut handle = null_mut();
        check(unsafe { bindgen::Plaintext_Create1(null_mut(), &mut handle) })?;
        Ok(Self { handle })
    }
}

impl RelinearizationKeys {
    pub(crate) fn empty() -> Result<Self> {
        let mut handle = null_mut();
        check(unsafe { bindgen::KSwitchKeys_Create1(&mut handle) })?;
        Ok(Self { handle })
    }
}

impl GaloisKeys {
    pub fn new() -> Result<Self> {
        let mut handle = null_mut();
        check(unsafe { bindgen::KSwitchKeys_Create1(&mut handle) })?;
        Ok(Self { handle })
    }
}
