// Links libmmcmc.so (built by `make -C mini_mcmc_amd/csrc`).  MMCMC_LIB_DIR names the directory that holds it;
// default: ../../mini_mcmc_amd relative to this crate (the in-tree build).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("MMCMC_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../mini_mcmc_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=mmcmc");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=MMCMC_LIB_DIR");
}
