// Links libmi355deflate.so.  MI355_DEFLATE_LIB_DIR = the directory that holds it
// (deflate-rs_amd/ of the framework tree, after `make -C deflate-rs_amd`).
use std::env;

fn main() {
    println!("cargo:rerun-if-env-changed=MI355_DEFLATE_LIB_DIR");
    if let Ok(dir) = env::var("MI355_DEFLATE_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=mi355deflate");
}
