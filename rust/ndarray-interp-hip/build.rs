// Links libndinterp_hip.so (built by `make -C ndarray-interp_amd/csrc`, hipcc --offload-arch=gfx950).
// NDINTERP_HIP_LIB_DIR names the directory that holds it; the HIP runtime it needs comes from ROCM_PATH.
fn main() {
    let dir = std::env::var("NDINTERP_HIP_LIB_DIR")
        .expect("set NDINTERP_HIP_LIB_DIR to the directory that holds libndinterp_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=ndinterp_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    let rocm = std::env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".into());
    println!("cargo:rustc-link-search=native={rocm}/lib");
    println!("cargo:rustc-link-lib=dylib=amdhip64"); // hipStreamPerThread / events used by the ring consumers
    println!("cargo:rerun-if-env-changed=NDINTERP_HIP_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
}
