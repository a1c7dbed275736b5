// Links librrt.so.  RRT_LIB_DIR = the directory that holds it (<repo>/rust-raytracer_amd); at run time the loader also
// needs /opt/rocm/lib (LD_LIBRARY_PATH or an rpath).
fn main() {
    let dir = std::env::var("RRT_LIB_DIR").unwrap_or_else(|_| "../rust-raytracer_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rrt");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=RRT_LIB_DIR");
}
