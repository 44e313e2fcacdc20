// Links liblasgun_hip.so (built by `make -C lasgun_amd/csrc` at the repository root: hipcc --offload-arch=gfx950).
// LASGUN_HIP_LIB_DIR overrides the directory the library is looked up in; by default it is <repo>/lasgun_amd.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("LASGUN_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
        manifest.join("../../../lasgun_amd") // bindings/rust/lasgun-hip-sys -> repository root
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=lasgun_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=LASGUN_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../../../include/lasgun_hip.h");
}
