// UNCOMPILED here (no Rust toolchain in the authoring image).
// Links libivp_hip.so; IVP_HIP_LIB_DIR names the directory that holds it (default: ../../ivp_amd of this repository,
// where `make -C ivp_amd/csrc` puts it).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("IVP_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../ivp_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=ivp_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=IVP_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/ivp_hip.h");
}
