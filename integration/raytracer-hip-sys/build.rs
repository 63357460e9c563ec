// Finds librt_hip.so.  RT_HIP_LIB_DIR names the directory that holds it (the `raytracer_amd/`
// directory of the MI355X tree after `python -c "import __graft_entry__ as g; g.build()"`);
// without it the system library path is searched.  The directory is also baked in as an rpath so
// that `cargo run` works without LD_LIBRARY_PATH.
use std::env;

fn main() {
    println!("cargo:rerun-if-env-changed=RT_HIP_LIB_DIR");
    if let Ok(dir) = env::var("RT_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=rt_hip");
}
