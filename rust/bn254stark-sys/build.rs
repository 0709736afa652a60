// Links libbn254stark.so (built by `make -C plonky2_bn254_amd/csrc`); BN254STARK_LIB_DIR overrides the search path.
fn main() {
    let dir = std::env::var("BN254STARK_LIB_DIR").unwrap_or_else(|_| "../../plonky2_bn254_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=bn254stark");
    println!("cargo:rerun-if-env-changed=BN254STARK_LIB_DIR");
}
