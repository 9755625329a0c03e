// Links librtx_hip.so.  RTX_LIB_DIR = the directory that holds it (rust-raytracer_amd/ after
// `make -C rust-raytracer_amd/csrc`); the run-time search path is set too, so `cargo run` finds it.
fn main() {
    let dir = std::env::var("RTX_LIB_DIR").expect("set RTX_LIB_DIR to the directory that holds librtx_hip.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rtx_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=RTX_LIB_DIR");
}
