// Points the linker at libzksaas_hip.so: ZKSAAS_LIB_DIR, or <repo>/zk-saas_amd next to this crate.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("ZKSAAS_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../zk-saas_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=zksaas_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=ZKSAAS_LIB_DIR");
}
