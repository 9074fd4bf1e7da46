//! The GPU batch executor seam: one graph node over `count` independent ciphertexts per call.
//!
//! `sunscreen_runtime/src/run.rs:160-341` issues one FFI call (and at least one allocation) per node per ciphertext.
//! Here a ciphertext batch is a device buffer `u64[count][size][K][N]` that never leaves HBM between nodes; operations
//! are asynchronous on the caller's HIP stream.  Transparent results (SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT,
//! `sunscreen/tests/features.rs:8-34`) are recorded on the device and surface at [`BatchEvaluator::check`] /
//! [`Program::run`], which is where this path synchronises.
use std::ffi::c_void;
use std::ptr::null_mut;

use crate::{bindgen, check, BFVEvaluator, GaloisKeys, RelinearizationKeys, Result};

/// A borrowed device buffer of `count` ciphertexts of `size` polynomials (memory is owned by the caller's allocator:
/// hipMalloc, a torch tensor, ...).  The library cannot check a device address: constructing one is the `unsafe` step, the
/// operations on a constructed batch are safe.
#[derive(Clone, Copy)]
pub struct DeviceBatch {
    ptr: *mut u64,
    size: u64,
    count: u64,
}

impl DeviceBatch {
    /// # Safety
    /// `ptr` must be a device address of at least `count * size * K * N` u64 words on the evaluator's device, valid (and,
    /// for outputs, not aliased by a concurrently running operation) until the stream work that uses it has completed.
    pub unsafe fn new(ptr: *mut u64, size: u64, count: u64) -> Self {
        Self { ptr, size, count }
    }
    pub fn ptr(&self) -> *mut u64 {
        self.ptr
    }
    pub fn size(&self) -> u64 {
        self.size
    }
    pub fn count(&self) -> u64 {
        self.count
    }
}

fn same_count(a: &DeviceBatch, b: &DeviceBatch) -> Result<()> {
    if a.count != b.count {
        return Err(crate::Error::InvalidArgument(format!("batches of {} and {} ciphertexts", a.count, b.count)));
    }
    Ok(())
}

pub struct BatchEvaluator<'e> {
    eval: &'e BFVEvaluator,
    stream: *mut c_void,
}

impl<'e> BatchEvaluator<'e> {
    /// `stream`: a `hipStream_t` (null = the default stream).
    pub fn new(eval: &'e BFVEvaluator, stream: *mut c_void) -> Self {
        Self { eval, stream }
    }
    fn h(&self) -> *mut c_void {
        self.eval.get_handle()
    }

    pub fn multiply_relin(&self, a: DeviceBatch, b: DeviceBatch, rk: &RelinearizationKeys, out: DeviceBatch) -> Result<()> {
        same_count(&a, &b)?;
        same_count(&a, &out)?;
        check(unsafe { bindgen::hipbfv_batch_multiply_relin(self.h(), a.ptr, b.ptr, rk.get_handle(), out.ptr, a.count, self.stream) })
    }
    pub fn multiply(&self, a: DeviceBatch, b: DeviceBatch, out: DeviceBatch) -> Result<()> {
        same_count(&a, &b)?;
        same_count(&a, &out)?;
        check(unsafe { bindgen::hipbfv_batch_multiply(self.h(), a.ptr, a.size, b.ptr, b.size, out.ptr, a.count, self.stream) })
    }
    pub fn relinearize(&self, ct3: DeviceBatch, rk: &RelinearizationKeys, out: DeviceBatch) -> Result<()> {
        check(unsafe { bindgen::hipbfv_batch_relinearize(self.h(), ct3.ptr, rk.get_handle(), out.ptr, ct3.count, self.stream) })
    }
    pub fn rotate_rows(&self, a: DeviceBatch, steps: i32, gk: &GaloisKeys, out: DeviceBatch) -> Result<()> {
        check(unsafe { bindgen::hipbfv_batch_rotate_rows(self.h(), a.ptr, steps, gk.get_handle(), out.ptr, a.count, self.stream) })
    }
    pub fn rotate_columns(&self, a: DeviceBatch, gk: &GaloisKeys, out: DeviceBatch) -> Result<()> {
        check(unsafe { bindgen::hipbfv_batch_rotate_columns(self.h(), a.ptr, gk.get_handle(), out.ptr, a.count, self.stream) })
    }
    // ---- per-key batches ----
    // The reference hands the keys over with every call (`sunscreen_runtime/src/run.rs:100-105`: `relin_keys:
    // &Option<&RelinearizationKeys>`, `galois_keys: &Option<&GaloisKeys>`; `runtime.rs:310-327`): a server that batches the
    // calls of many clients holds one key set per client.  Item i uses `keys[key_index[i]]`; items need not be grouped by key.
    fn key_index_ok(key_index: &[u32], sets: usize, count: u64) -> Result<()> {
        if key_index.len() as u64 != count || key_index.iter().any(|&k| k as usize >= sets) {
            return Err(crate::Error::InvalidArgument(format!("{} key indices for {} items over {} key sets", key_index.len(), count, sets)));
        }
        Ok(())
    }
    pub fn multiply_relin_keys(&self, a: DeviceBatch, b: DeviceBatch, keys: &[&RelinearizationKeys], key_index: &[u32], out: DeviceBatch) -> Result<()> {
        same_count(&a, &b)?;
        same_count(&a, &out)?;
        Self::key_index_ok(key_index, keys.len(), a.count)?;
        let handles: Vec<*mut c_void> = keys.iter().map(|k| k.get_handle()).collect();
        check(unsafe {
            bindgen::hipbfv_batch_multiply_relin_keys(self.h(), a.ptr, b.ptr, handles.as_ptr(), handles.len() as u64, key_index.as_ptr(), out.ptr, a.count, self.stream)
        })
    }
    pub fn relinearize_keys(&self, ct3: DeviceBatch, keys: &[&RelinearizationKeys], key_index: &[u32], out: DeviceBatch) -> Result<()> {
        same_count(&ct3, &out)?;
        Self::key_index_ok(key_index, keys.len(), ct3.count)?;
        let handles: Vec<*mut c_void> = keys.iter().map(|k| k.get_handle()).collect();
        check(unsafe {
            bindgen::hipbfv_batch_relinearize_keys(self.h(), ct3.ptr, handles.as_ptr(), handles.len() as u64, key_index.as_ptr(), out.ptr, ct3.count, self.stream)
        })
    }
    pub fn rotate_rows_keys(&self, a: DeviceBatch, steps: i32, keys: &[&GaloisKeys], key_index: &[u32], out: DeviceBatch) -> Result<()> {
        same_count(&a, &out)?;
        Self::key_index_ok(key_index, keys.len(), a.count)?;
        let handles: Vec<*mut c_void> = keys.iter().map(|k| k.get_handle()).collect();
        check(unsafe {
            bindgen::hipbfv_batch_rotate_rows_keys(self.h(), a.ptr, steps, handles.as_ptr(), handles.len() as u64, key_index.as_ptr(), out.ptr, a.count, self.stream)
        })
    }
    pub fn rotate_columns_keys(&self, a: DeviceBatch, keys: &[&GaloisKeys], key_index: &[u32], out: DeviceBatch) -> Result<()> {
        same_count(&a, &out)?;
        Self::key_index_ok(key_index, keys.len(), a.count)?;
        let handles: Vec<*mut c_void> = keys.iter().map(|k| k.get_handle()).collect();
        check(unsafe {
            bindgen::hipbfv_batch_rotate_columns_keys(self.h(), a.ptr, handles.as_ptr(), handles.len() as u64, key_index.as_ptr(), out.ptr, a.count, self.stream)
        })
    }
    pub fn add(&self, a: DeviceBatch, b: DeviceBatch, out: DeviceBatch) -> Result<()> {
        same_count(&a, &b)?;
        same_count(&a, &out)?;
        check(unsafe { bindgen::hipbfv_batch_add(self.h(), a.ptr, b.ptr, out.ptr, a.size, a.count, self.stream) })
    }
    pub fn sub(&self, a: DeviceBatch, b: DeviceBatch, out: DeviceBatch) -> Result<()> {
        same_count(&a, &b)?;
        same_count(&a, &out)?;
        check(unsafe { bindgen::hipbfv_batch_sub(self.h(), a.ptr, b.ptr, out.ptr, a.size, a.count, self.stream) })
    }
    pub fn negate(&self, a: DeviceBatch, out: DeviceBatch) -> Result<()> {
        check(unsafe { bindgen::hipbfv_batch_negate(self.h(), a.ptr, out.ptr, a.size, a.count, self.stream) })
    }
    /// Lift `count` coefficient-form plaintexts (`u64[count][N]`, stride `plain_stride` words) to the data primes and transform
    /// them: `pntt` = `u64[count][K][N]`, the operand `Input::PlaintextsNtt` takes.  An all-zero plaintext has no transformed
    /// form -- SEAL refuses every product with it (`sunscreen/tests/features.rs:8-34`) and the consumers cannot see it any
    /// more -- so this call ends with [`BatchEvaluator::check`] and fails with the index of the first one.
    ///
    /// # Safety
    /// `plain` and `pntt` must be device addresses of `count` plaintexts / `count * K * N` words on the evaluator's device.
    pub unsafe fn plain_to_ntt(&self, plain: *const u64, plain_stride: u64, pntt: *mut u64, count: u64) -> Result<()> {
        check(bindgen::hipbfv_batch_plain_to_ntt(self.h(), plain, plain_stride, pntt, count, self.stream))?;
        self.check()
    }
    /// Synchronise the stream; `Err(InternalError(COR_E_INVALIDOPERATION, ..))` if any operation since the last call
    /// produced a transparent ciphertext.
    pub fn check(&self) -> Result<()> {
        let mut first = 0u64;
        check(unsafe { bindgen::hipbfv_batch_status(self.h(), &mut first, self.stream) })
    }
}

/// A compiled `FheProgram` graph (serde JSON of `sunscreen_fhe_program::FheProgram`, or built node by node) executed over
/// a batch of independent input sets: the replacement of `run_program_unchecked` (`run.rs:100-357`).
pub struct Program {
    handle: *mut c_void,
}
unsafe impl Sync for Program {}
unsafe impl Send for Program {}

pub enum Input {
    /// `u64[batch][2][K][N]` on the device
    Ciphertexts(*const u64),
    /// per-item plaintexts `u64[batch][N]` (stride N) or one shared `u64[N]` (stride 0), coefficient form, on the device
    Plaintexts { ptr: *const u64, stride: u64 },
    /// plaintexts already in transform form, `u64[batch][K][N]` (stride K*N) or shared (stride 0): the output of
    /// `hipbfv_batch_plain_to_ntt` -- a server's static database (examples/pir) is transformed once, not per query
    PlaintextsNtt { ptr: *const u64, stride: u64 },
}

impl Program {
    pub fn from_json(json: &str) -> Result<Self> {
        let mut handle = null_mut();
        check(unsafe { bindgen::hipbfv_Program_Create(&mut handle) })?;
        let p = Self { handle };
        check(unsafe { bindgen::hipbfv_Program_LoadJson(p.handle, json.as_ptr() as *const _, json.len() as u64) })?;
        Ok(p)
    }

    /// One output buffer `u64[batch][2][K][N]` per `OutputCiphertext` node, in node order.
    ///
    /// # Safety
    /// Every pointer in `inputs` and `outputs` must be a device address of the size its kind implies for `batch` items on
    /// the evaluator's device, valid until `stream` has drained (this call synchronises it before returning).
    pub unsafe fn run(
        &self, eval: &BFVEvaluator, batch: u64, inputs: &[Input], rk: Option<&RelinearizationKeys>, gk: Option<&GaloisKeys>,
        outputs: &[*mut u64], stream: *mut c_void,
    ) -> Result<()> {
        let kinds: Vec<u32> = inputs.iter().map(|i| match i { Input::Ciphertexts(_) => 0, Input::Plaintexts { .. } => 1, Input::PlaintextsNtt { .. } => 2 }).collect();
        let ptrs: Vec<*const u64> = inputs
            .iter()
            .map(|i| match i { Input::Ciphertexts(p) => *p, Input::Plaintexts { ptr, .. } | Input::PlaintextsNtt { ptr, .. } => *ptr })
            .collect();
        let strides: Vec<u64> = inputs
            .iter()
            .map(|i| match i { Input::Ciphertexts(_) => 0, Input::Plaintexts { stride, .. } | Input::PlaintextsNtt { stride, .. } => *stride })
            .collect();
        check(bindgen::hipbfv_Program_Run(
            self.handle, eval.get_handle(), batch, inputs.len() as u64, kinds.as_ptr(), ptrs.as_ptr(), strides.as_ptr(),
            rk.map_or(null_mut(), |k| k.get_handle()), gk.map_or(null_mut(), |k| k.get_handle()),
            outputs.len() as u64, outputs.as_ptr(), stream,
        ))
    }
}

impl Program {
    /// [`Program::run`] for a batch whose input sets belong to several clients: input set i runs with `rk[key_index[i]]`,
    /// `gk[key_index[i]]` (an entry may be `None` when the program needs no such key) -- the reference's per-call keys
    /// (`run.rs:100-105`) kept per input set.
    ///
    /// # Safety
    /// As for [`Program::run`].
    pub unsafe fn run_keys(
        &self, eval: &BFVEvaluator, batch: u64, inputs: &[Input], rk: &[Option<&RelinearizationKeys>], gk: &[Option<&GaloisKeys>],
        key_index: &[u32], outputs: &[*mut u64], stream: *mut c_void,
    ) -> Result<()> {
        let sets = rk.len().max(gk.len());
        if key_index.len() as u64 != batch || key_index.iter().any(|&k| k as usize >= sets) {
            return Err(crate::Error::InvalidArgument(format!("{} key indices for {} input sets over {} key sets", key_index.len(), batch, sets)));
        }
        let kinds: Vec<u32> = inputs.iter().map(|i| match i { Input::Ciphertexts(_) => 0, Input::Plaintexts { .. } => 1, Input::PlaintextsNtt { .. } => 2 }).collect();
        let ptrs: Vec<*const u64> = inputs
            .iter()
            .map(|i| match i { Input::Ciphertexts(p) => *p, Input::Plaintexts { ptr, .. } | Input::PlaintextsNtt { ptr, .. } => *ptr })
            .collect();
        let strides: Vec<u64> = inputs
            .iter()
            .map(|i| match i { Input::Ciphertexts(_) => 0, Input::Plaintexts { stride, .. } | Input::PlaintextsNtt { stride, .. } => *stride })
            .collect();
        let rks: Vec<*mut c_void> = (0..sets).map(|i| rk.get(i).copied().flatten().map_or(null_mut(), |k| k.get_handle())).collect();
        let gks: Vec<*mut c_void> = (0..sets).map(|i| gk.get(i).copied().flatten().map_or(null_mut(), |k| k.get_handle())).collect();
        check(bindgen::hipbfv_Program_RunKeys(
            self.handle, eval.get_handle(), batch, inputs.len() as u64, kinds.as_ptr(), ptrs.as_ptr(), strides.as_ptr(),
            sets as u64, rks.as_ptr(), gks.as_ptr(), key_index.as_ptr(), outputs.len() as u64, outputs.as_ptr(), stream,
        ))
    }
}

impl Drop for Program {
    fn drop(&mut self) {
        check(unsafe { bindgen::hipbfv_Program_Destroy(self.handle) }).expect("hipbfv_Program_Destroy");
    }
}
