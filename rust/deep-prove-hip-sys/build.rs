// links libdeepprove_hip.so from DEEP_PROVE_HIP_LIB_DIR (the directory deep-prove_amd/ of the repository after __graft_entry__.build())
fn main() {
    if let Ok(dir) = std::env::var("DEEP_PROVE_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=deepprove_hip");
    println!("cargo:rerun-if-env-changed=DEEP_PROVE_HIP_LIB_DIR");
}
