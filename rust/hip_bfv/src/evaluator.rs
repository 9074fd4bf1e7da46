//! The trait `sunscreen_runtime` programs against, and its MI355X implementation.
//!
//! Method names, argument order and in-place / out-of-place pairing follow `seal_fhe::Evaluator`
//! (`seal_fhe/src/evaluator.rs:7-280`) because that is the contract; the bodies are one C call each.  The runtime uses
//! ten of the methods (`run.rs:185-308`): rotate_rows, add, add_plain, multiply, multiply_plain, rotate_columns,
//! relinearize, negate, sub, sub_plain.
use std::ffi::c_void;
use std::ptr::null_mut;

use crate::{bindgen, check, Ciphertext, Context, GaloisKeys, Plaintext, RelinearizationKeys, Result};

pub trait Evaluator {
    type Plaintext;
    type Ciphertext;

    fn negate_inplace(&self, a: &mut Self::Ciphertext) -> Result<()>;
    fn negate(&self, a: &Self::Ciphertext) -> Result<Self::Ciphertext>;
    fn add_inplace(&self, a: &mut Self::Ciphertext, b: &Self::Ciphertext) -> Result<()>;
    fn add(&self, a: &Self::Ciphertext, b: &Self::Ciphertext) -> Result<Self::Ciphertext>;
    fn add_many(&self, a: &[Self::Ciphertext]) -> Result<Self::Ciphertext>;
    fn multiply_many(&self, a: &[Self::Ciphertext], relin_keys: &RelinearizationKeys) -> Result<Self::Ciphertext>;
    fn sub_inplace(&self, a: &mut Self::Ciphertext, b: &Self::Ciphertext) -> Result<()>;
    fn sub(&self, a: &Self::Ciphertext, b: &Self::Ciphertext) -> Result<Self::Ciphertext>;
    fn multiply_inplace(&self, a: &mut Self::Ciphertext, b: &Self::Ciphertext) -> Result<()>;
    fn multiply(&self, a: &Self::Ciphertext, b: &Self::Ciphertext) -> Result<Self::Ciphertext>;
    fn square_inplace(&self, a: &mut Self::Ciphertext) -> Result<()>;
    fn square(&self, a: &Self::Ciphertext) -> Result<Self::Ciphertext>;
    fn mod_switch_to_next(&self, a: &Self::Ciphertext) -> Result<Self::Ciphertext>;
    fn mod_switch_to_next_inplace(&self, a: &Self::Ciphertext) -> Result<()>;
    fn mod_switch_to_next_plaintext(&self, a: &Self::Plaintext) -> Result<Self::Plaintext>;
    fn mod_switch_to_next_inplace_plaintext(&self, a: &Self::Plaintext) -> Result<()>;
    fn exponentiate(&self, a: &Self::Ciphertext, exponent: u64, relin_keys: &RelinearizationKeys) -> Result<Self::Ciphertext>;
    fn exponentiate_inplace(&self, a: &Self::Ciphertext, exponent: u64, relin_keys: &RelinearizationKeys) -> Result<()>;
    fn add_plain(&self, a: &Self::Ciphertext, b: &Self::Plaintext) -> Result<Self::Ciphertext>;
    fn add_plain_inplace(&self, a: &mut Self::Ciphertext, b: &Self::Plaintext) -> Result<()>;
    fn sub_plain(&self, a: &Self::Ciphertext, b: &Self::Plaintext) -> Result<Self::Ciphertext>;
    fn sub_plain_inplace(&self, a: &mut Self::Ciphertext, b: &Self::Plaintext) -> Result<()>;
    fn multiply_plain(&self, a: &Self::Ciphertext, b: &Self::Plaintext) -> Result<Self::Ciphertext>;
    fn multiply_plain_inplace(&self, a: &mut Self::Ciphertext, b: &Self::Plaintext) -> Result<()>;
    fn relinearize_inplace(&self, a: &mut Self::Ciphertext, relin_keys: &RelinearizationKeys) -> Result<()>;
    fn relinearize(&self, a: &Self::Ciphertext, relin_keys: &RelinearizationKeys) -> Result<Self::Ciphertext>;
    fn rotate_rows(&self, a: &Self::Ciphertext, steps: i32, galois_keys: &GaloisKeys) -> Result<Self::Ciphertext>;
    fn rotate_rows_inplace(&self, a: &Self::Ciphertext, steps: i32, galois_keys: &GaloisKeys) -> Result<()>;
    fn rotate_columns(&self, a: &Self::Ciphertext, galois_keys: &GaloisKeys) -> Result<Self::Ciphertext>;
    fn rotate_columns_inplace(&self, a: &Self::Ciphertext, galois_keys: &GaloisKeys) -> Result<()>;
}

/// `BFVEvaluator::new(&ctx)` as in `seal_fhe/src/bfv_evaluator.rs:27-29`.  One evaluator is shared by reference across
/// the runtime's rayon workers (`run.rs:415-469`); libhipbfv gives every host thread its own HIP stream.
pub struct BFVEvaluator {
    handle: *mut c_void,
}
unsafe impl Sync for BFVEvaluator {}
unsafe impl Send for BFVEvaluator {}

impl BFVEvaluator {
    pub fn new(ctx: &Context) -> Result<Self> {
        let mut handle = null_mut();
        check(unsafe { bindgen::Evaluator_Create(ctx.get_handle(), &mut handle) })?;
        Ok(Self { handle })
    }
    pub fn get_handle(&self) -> *mut c_void {
        self.handle
    }
}

impl Drop for BFVEvaluator {
    fn drop(&mut self) {
        check(unsafe { bindgen::Evaluator_Destroy(self.handle) }).expect("Evaluator_Destroy");
    }
}

// out-of-place = fresh destination, callee resizes it; in-place = destination aliases the first operand
macro_rules! out_of_place {
    ($self:ident, |$dst:ident| $call:expr) => {{
        let out = Ciphertext::new()?;
        let $dst = out.get_handle();
        check(unsafe { $call })?;
        Ok(out)
    }};
}

impl Evaluator for BFVEvaluator {
    type Plaintext = Plaintext;
    type Ciphertext = Ciphertext;

    fn negate_inplace(&self, a: &mut Ciphertext) -> Result<()> {
        check(unsafe { bindgen::Evaluator_Negate(self.handle, a.get_handle(), a.get_handle()) })
    }
    fn negate(&self, a: &Ciphertext) -> Result<Ciphertext> {
        out_of_place!(self, |d| bindgen::Evaluator_Negate(self.handle, a.get_handle(), d))
    }
    fn add_inplace(&self, a: &mut Ciphertext, b: &Ciphertext) -> Result<()> {
        check(unsafe { bindgen::Evaluator_Add(self.handle, a.get_handle(), b.get_handle(), a.get_handle()) })
    }
    fn add(&self, a: &Ciphertext, b: &Ciphertext) -> Result<Ciphertext> {
        out_of_place!(self, |d| bindgen::Evaluator_Add(self.handle, a.get_handle(), b.get_handle(), d))
    }
    fn add_many(&self, a: &[Ciphertext]) -> Result<Ciphertext> {
        let mut list: Vec<*mut c_void> = a.iter().map(|c| c.get_handle()).collect();
        out_of_place!(self, |d| bindgen::Evaluator_AddMany(self.handle, list.len() as u64, list.as_mut_ptr(), d))
    }
    fn multiply_many(&self, a: &[Ciphertext], relin_keys: &RelinearizationKeys) -> Result<Ciphertext> {
        let mut list: Vec<*mut c_void> = a.iter().map(|c| c.get_handle()).collect();
        out_of_place!(self, |d| bindgen::Evaluator_MultiplyMany(self.handle, list.len() as u64, list.as_mut_ptr(), relin_keys.get_handle(), d, null_mut()))
    }
    fn sub_inplace(&self, a: &mut Ciphertext, b: &Ciphertext) -> Result<()> {
        check(unsafe { bindgen::Evaluator_Sub(self.handle, a.get_handle(), b.get_handle(), a.get_handle()) })
    }
    fn sub(&self, a: &Ciphertext, b: &Ciphertext) -> Result<Ciphertext> {
        out_of_place!(self, |d| bindgen::Evaluator_Sub(self.handle, a.get_handle(), b.get_handle(), d))
    }
    fn multiply_inplace(&self, a: &mut Ciphertext, b: &Ciphertext) -> Result<()> {
        check(unsafe { bindgen::Evaluator_Multiply(self.handle, a.get_handle(), b.get_handle(), a.get_handle(), null_mut()) })
    }
    fn multiply(&self, a: &Ciphertext, b: &Ciphertext) -> Result<Ciphertext> {
        out_of_place!(self, |d| bindgen::Evaluator_Multiply(self.handle, a.get_handle(), b.get_handle(), d, null_mut()))
    }
    fn square_inplace(&self, a: &mut Ciphertext) -> Result<()> {
        check(unsafe { bindgen::Evaluator_Square(self.handle, a.get_handle(), a.get_handle(), null_mut()) })
    }
    fn square(&self, a: &Ciphertext) -> Result<Ciphertext> {
        out_of_place!(self, |d| bindgen::Evaluator_Square(self.handle, a.get_handle(), d, null_mut()))
    }
    fn mod_switch_to_next(&self, a: &Ciphertext) -> Result<Ciphertext> {
        out_of_place!(self, |d| bindgen::Evaluator_ModSwitchToNext1(self.handle, a.get_handle(), d, null_mut()))
    }
    fn mod_switch_to_next_inplace(&self, a: &Ciphertext) -> Result<()> {
        check(unsafe { bindgen::Evaluator_ModSwitchToNext1(self.handle, a.get_handle(), a.get_handle(), null_mut()) })
    }
    fn mod_switch_to_next_plaintext(&self, a: &Plaintext) -> Result<Plaintext> {
        let out = Plaintext::new()?;
        check(unsafe { bindgen::Evaluator_ModSwitchToNext2(self.handle, a.get_handle(), out.get_handle()) })?;
        Ok(out)
    }
    fn mod_switch_to_next_inplace_plaintext(&self, a: &Plaintext) -> Result<()> {
        check(unsafe { bindgen::Evaluator_ModSwitchToNext2(self.handle, a.get_handle(), a.get_handle()) })
    }
    fn exponentiate(&self, a: &Ciphertext, exponent: u64, relin_keys: &RelinearizationKeys) -> Result<Ciphertext> {
        out_of_place!(self, |d| bindgen::Evaluator_Exponentiate(self.handle, a.get_handle(), exponent, relin_keys.get_handle(), d, null_mut()))
    }
    fn exponentiate_inplace(&self, a: &Ciphertext, exponent: u64, relin_keys: &RelinearizationKeys) -> Result<()> {
        check(unsafe { bindgen::Evaluator_Exponentiate(self.handle, a.get_handle(), exponent, relin_keys.get_handle(), a.get_handle(), null_mut()) })
    }
    fn add_plain(&self, a: &Ciphertext, b: &Plaintext) -> Result<Ciphertext> {
        out_of_place!(self, |d| bindgen::Evaluator_AddPlain(self.handle, a.get_handle(), b.get_handle(), d))
    }
    fn add_plain_inplace(&self, a: &mut Ciphertext, b: &Plaintext) -> Result<()> {
        check(unsafe { bindgen::Evaluator_AddPlain(self.handle, a.get_handle(), b.get_handle(), a.get_handle()) })
    }
    fn sub_plain(&self, a: &Ciphertext, b: &Plaintext) -> Result<Ciphertext> {
        out_of_place!(self, |d| bindgen::Evaluator_SubPlain(self.handle, a.get_handle(), b.get_handle(), d))
    }
    fn sub_plain_inplace(&self, a: &mut Ciphertext, b: &Plaintext) -> Result<()> {
        check(unsafe { bindgen::Evaluator_SubPlain(self.handle, a.get_handle(), b.get_handle(), a.get_handle()) })
    }
    fn multiply_plain(&self, a: &Ciphertext, b: &Plaintext) -> Result<Ciphertext> {
        out_of_place!(self, |d| bindgen::Evaluator_MultiplyPlain(self.handle, a.get_handle(), b.get_handle(), d, null_mut()))
    }
    fn multiply_plain_inplace(&self, a: &mut Ciphertext, b: &Plaintext) -> Result<()> {
        check(unsafe { bindgen::Evaluator_MultiplyPlain(self.handle, a.get_handle(), b.get_handle(), a.get_handle(), null_mut()) })
    }
    fn relinearize_inplace(&self, a: &mut Ciphertext, relin_keys: &RelinearizationKeys) -> Result<()> {
        check(unsafe { bindgen::Evaluator_Relinearize(self.handle, a.get_handle(), relin_keys.get_handle(), a.get_handle(), null_mut()) })
    }
    fn relinearize(&self, a: &Ciphertext, relin_keys: &RelinearizationKeys) -> Result<Ciphertext> {
        out_of_place!(self, |d| bindgen::Evaluator_Relinearize(self.handle, a.get_handle(), relin_keys.get_handle(), d, null_mut()))
    }
    fn rotate_rows(&self, a: &Ciphertext, steps: i32, galois_keys: &GaloisKeys) -> Result<Ciphertext> {
        out_of_place!(self, |d| bindgen::Evaluator_RotateRows(self.handle, a.get_handle(), steps, galois_keys.get_handle(), d, null_mut()))
    }
    fn rotate_rows_inplace(&self, a: &Ciphertext, steps: i32, galois_keys: &GaloisKeys) -> Result<()> {
        check(unsafe { bindgen::Evaluator_RotateRows(self.handle, a.get_handle(), steps, galois_keys.get_handle(), a.get_handle(), null_mut()) })
    }
    fn rotate_columns(&self, a: &Ciphertext, galois_keys: &GaloisKeys) -> Result<Ciphertext> {
        out_of_place!(self, |d| bindgen::Evaluator_RotateColumns(self.handle, a.get_handle(), galois_keys.get_handle(), d, null_mut()))
    }
    fn rotate_columns_inplace(&self, a: &Ciphertext, galois_keys: &GaloisKeys) -> Result<()> {
        check(unsafe { bindgen::Evaluator_RotateColumns(self.handle, a.get_handle(), galois_keys.get_handle(), a.get_handle(), null_mut()) })
    }
}
