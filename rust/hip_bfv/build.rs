// rust/hip_bfv/build.rs -- bind include/hipbfv.h and link libhipbfv.so.
//
// The reference crate's build script cmake-builds SEAL and runs bindgen over seal/c/*.h with an allow-list of symbol
// families (seal_fhe/build.rs:44-66, 157-180).  Here there is nothing to build on the Rust side: the library is built by
// `make -C sunscreen_amd/csrc` (hipcc, gfx950); this script only generates the declarations and emits the link flags.
//
//   HIPBFV_ROOT   checkout of this repository (default: two levels above this crate)
use std::{env, path::PathBuf};

fn main() {
    let root = env::var("HIPBFV_ROOT")
        .map(PathBuf::from)
        .unwrap_or_else(|_| PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../.."));
    let header = root.join("include/hipbfv.h");
    let libdir = root.join("sunscreen_amd/lib");
    println!("cargo:rerun-if-changed={}", header.display());
    println!("cargo:rustc-link-search=native={}", libdir.display());
    println!("cargo:rustc-link-lib=dylib=hipbfv");
    // consumers find the shared object at run time without LD_LIBRARY_PATH
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", libdir.display());

    let bindings = bindgen::Builder::default()
        .header(header.to_str().unwrap())
        // the SEAL-named families seal_fhe binds (tests/golden/seal_fhe_ffi_symbols.txt lists all 121 functions) ...
        .allowlist_function("(Evaluator|Ciphertext|Plaintext|KSwitchKeys|SecretKey|PublicKey|KeyGenerator)_.*")
        .allowlist_function("(BatchEncoder|Encryptor|Decryptor|PolynomialArray|SEALContext|EncParams|Modulus|CoeffModulus)_.*")
        // ... and the library's own extensions: raw arrays, batched device-pointer entry points, program graphs
        .allowlist_function("hipbfv_.*")
        .allowlist_var("HIPBFV_.*")
        .generate()
        .expect("bindgen over include/hipbfv.h");
    bindings
        .write_to_file(PathBuf::from(env::var("OUT_DIR").unwrap()).join("bindings.rs"))
        .expect("write bindings.rs");
}
