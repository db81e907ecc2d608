// Links libbzhip.so.  Set BZHIP_LIB_DIR to the directory holding it (default: ../banzai_amd).
fn main() {
    let dir = std::env::var("BZHIP_LIB_DIR").unwrap_or_else(|_| "../banzai_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=bzhip");
    println!("cargo:rerun-if-env-changed=BZHIP_LIB_DIR");
}
