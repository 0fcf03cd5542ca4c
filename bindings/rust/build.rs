// Links against the prebuilt librtmi.so (built by `python -m raytracing_rust_amd.build`).
// RTMI_LIB_DIR overrides the default in-tree location.
fn main() {
    let dir = std::env::var("RTMI_LIB_DIR")
        .unwrap_or_else(|_| format!("{}/../../raytracing_rust_amd/lib", env!("CARGO_MANIFEST_DIR")));
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rtmi");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=RTMI_LIB_DIR");
}
