// `halo2_proofs/build.rs` of the patched halo2-axiom checkout: link libzkhip.so (built by `python __graft_entry__.py`, or
// `make -C zksnap_circuits_halo2_amd/csrc`, into zksnap_circuits_halo2_amd/).  ZKHIP_LIB_DIR names that directory.
fn main() {
    println!("cargo:rerun-if-env-changed=ZKHIP_LIB_DIR");
    let dir = std::env::var("ZKHIP_LIB_DIR").expect("set ZKHIP_LIB_DIR to the directory that holds libzkhip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=zkhip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
}
