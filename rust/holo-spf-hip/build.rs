// Links libholo_spf_hip.so (built by `python -m holo_amd.build`, hipcc --offload-arch=gfx950).
// HOLO_SPF_HIP_DIR = the directory that holds the library (holo_amd/ of the engine repository).
use std::env;

fn main() {
    println!("cargo:rerun-if-env-changed=HOLO_SPF_HIP_DIR");
    if let Ok(dir) = env::var("HOLO_SPF_HIP_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=holo_spf_hip");
}
