// Links libmmcmc.so (built by `make -C mini_mcmc_amd/csrc`).  MMCMC_LIB_DIR names the directory that holds it;
// default: ../../mini_mcmc_amd relative to this crate (the in-tree build).
use std::env;
use std::path::PathBuf;
use std::process::Command;

fn main() {
    let dir = env::var("MMCMC_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../mini_mcmc_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=mmcmc");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=MMCMC_LIB_DIR");

    // src/lib.rs asserts the size, alignment and field offsets of every struct (generated from include/mmcmc.h by
    // tools/gen_rust_sys.py); layout_check.c asserts the same numbers against the header with the C compiler, so a header
    // that moved on without the crate being regenerated stops the build here instead of corrupting memory at run time.
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let check = manifest.join("layout_check.c");
    println!("cargo:rerun-if-changed={}", check.display());
    println!("cargo:rerun-if-changed={}", manifest.join("../../include/mmcmc.h").display());
    let cc = env::var("CC").unwrap_or_else(|_| "cc".to_string());
    match Command::new(&cc).args(["-std=c11", "-fsyntax-only"]).arg(&check).status() {
        Ok(s) if s.success() => {}
        Ok(_) => panic!("layout_check.c: include/mmcmc.h no longer has the layouts this crate was generated for -- run python tools/gen_rust_sys.py"),
        Err(_) => println!("cargo:warning=no C compiler ({}): the header's struct layouts were not re-checked", cc),
    }
}
