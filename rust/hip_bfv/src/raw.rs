//! Raw-array import / export: data crosses the boundary without serialising.  Layouts are SEAL's
//! (`seal_fhe/src/plaintext_ciphertext.rs:303-314`): ciphertext `u64[size][K][N]` canonical residues in coefficient form,
//! key-switching key `u64[K][2][K+1][N]` in NTT form with the special prime last.
//!
//! Every slice is checked against the context's sizes before its pointer crosses the boundary: the C side trusts the
//! length implied by (size, K, N), so a short slice would be read out of bounds from safe code.
use crate::{bindgen, check, Ciphertext, Context, Error, GaloisKeys, RelinearizationKeys, Result};

fn expect_len(what: &str, got: usize, want: u64) -> Result<()> {
    if got as u64 != want {
        return Err(Error::InvalidArgument(format!("{what}: {got} words given, {want} expected")));
    }
    Ok(())
}

/// Words of one key-switching key `u64[K][2][K+1][N]`.
fn kswitch_words(ctx: &Context) -> Result<u64> {
    let (n, k, kk, _) = ctx.info()?;
    Ok(k * 2 * kk * n)
}

impl Ciphertext {
    pub fn from_raw(ctx: &Context, size: usize, data: &[u64]) -> Result<Self> {
        let (n, k, _, _) = ctx.info()?;
        expect_len("ciphertext u64[size][K][N]", data.len(), size as u64 * k * n)?;
        let ct = Ciphertext::new()?;
        check(unsafe { bindgen::hipbfv_Ciphertext_Assign(ct.handle, ctx.handle, size as u64, data.as_ptr()) })?;
        Ok(ct)
    }

    pub fn to_raw(&self, poly_modulus_degree: usize) -> Result<Vec<u64>> {
        let words = (self.num_polynomials() * self.coeff_modulus_size()) as usize * poly_modulus_degree;
        let mut out = vec![0u64; words];
        check(unsafe { bindgen::hipbfv_Ciphertext_Export(self.handle, out.as_mut_ptr(), words as u64) })?;
        Ok(out)
    }

    /// Device address of the ciphertext's `u64[size][K][N]` (valid while `self` lives and is not written).
    pub fn device_ptr(&self) -> Result<*mut u64> {
        let mut p = std::ptr::null_mut();
        check(unsafe { bindgen::hipbfv_Ciphertext_DevicePtr(self.handle, &mut p) })?;
        Ok(p)
    }
}

impl RelinearizationKeys {
    /// SEAL `KSwitchKeys::data()[0]` of a relinearisation key.
    pub fn from_raw(ctx: &Context, key: &[u64]) -> Result<Self> {
        expect_len("relinearisation key u64[K][2][K+1][N]", key.len(), kswitch_words(ctx)?)?;
        let rk = RelinearizationKeys::empty()?;
        check(unsafe { bindgen::hipbfv_KSwitchKeys_AssignRelin(rk.handle, ctx.handle, key.as_ptr()) })?;
        Ok(rk)
    }
}

impl GaloisKeys {
    /// Key for the automorphism x -> x^galois_elt (index (galois_elt - 1) / 2 of SEAL's key list).
    pub fn insert_raw(&mut self, ctx: &Context, galois_elt: u32, key: &[u64]) -> Result<()> {
        expect_len("Galois key u64[K][2][K+1][N]", key.len(), kswitch_words(ctx)?)?;
        check(unsafe { bindgen::hipbfv_KSwitchKeys_AssignGalois(self.handle, ctx.handle, galois_elt, key.as_ptr()) })
    }
}
