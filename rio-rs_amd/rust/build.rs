// build.rs — link search path for the MI355X placement provider (feature `gpu`).
//
// Goes into the reference tree as `rio-rs/build.rs`.  With the feature off it does nothing.  With it on, the
// directory of librio_gp.so comes from RIO_GP_LIB_DIR (the library is built outside cargo, by hipcc: see
// INTEGRATION.md section 1); it is also written into the binary's rpath so that `cargo test --features gpu`
// finds the library without LD_LIBRARY_PATH.  NOT COMPILED IN THIS REPOSITORY (no cargo in the build image).
fn main() {
    println!("cargo:rerun-if-env-changed=RIO_GP_LIB_DIR");
    if std::env::var_os("CARGO_FEATURE_GPU").is_none() {
        return;
    }
    let dir = std::env::var("RIO_GP_LIB_DIR")
        .expect("feature `gpu`: set RIO_GP_LIB_DIR to the directory that holds librio_gp.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rio_gp");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    // libamdhip64 is a dependency of librio_gp.so itself; make the loader find it the same way
    let rocm = std::env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-arg=-Wl,-rpath,{rocm}/lib");
}
