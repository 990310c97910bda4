// build.rs — link hbetune against libhbegp.so (the MI355X GP engine, include/hbegp.h of the hbetune_rs_amd repository).
//
// NOT COMPILED IN THE REPOSITORY THAT SHIPS IT: its build image has no Rust toolchain (cargo / rustc absent).  This is
// the file a maintainer drops next to hbetune's Cargo.toml together with `src/core/gpr_gpu.rs` (integration/gpr_gpu.rs);
// it uses nothing beyond std, so it needs no build-dependencies.
//
//   HBEGP_LIB_DIR   directory that holds libhbegp.so   (default: ../hbetune_rs_amd/hbetune_rs_amd)
//   ROCM_PATH       ROCm installation                  (default: /opt/rocm) -- libamdhip64 is a dependency of libhbegp
//
// Cargo.toml additions:   [package] build = "build.rs"      [features] gpu = []
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=HBEGP_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
    println!("cargo:rerun-if-changed=build.rs");
    if env::var_os("CARGO_FEATURE_GPU").is_none() {
        return; // the CPU-only build does not know about libhbegp
    }
    let lib_dir = env::var_os("HBEGP_LIB_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|| PathBuf::from("../hbetune_rs_amd/hbetune_rs_amd"));
    let lib_dir = lib_dir.canonicalize().unwrap_or(lib_dir);
    if !lib_dir.join("libhbegp.so").exists() {
        panic!(
            "libhbegp.so not found in {}: build it (`make` in hbetune_rs_amd) or set HBEGP_LIB_DIR",
            lib_dir.display()
        );
    }
    let rocm = env::var_os("ROCM_PATH")
        .map(PathBuf::from)
        .unwrap_or_else(|| PathBuf::from("/opt/rocm"));
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-search=native={}", rocm.join("lib").display());
    println!("cargo:rustc-link-lib=dylib=hbegp");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    // run-time lookup without LD_LIBRARY_PATH (tests, `cargo run`)
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib_dir.display());
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", rocm.join("lib").display());
}
