// build.rs for the reference crate once it calls libmirt (not compiled in this repository's image).
fn main() {
    if let Ok(dir) = std::env::var("MIRT_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=mirt");
    println!("cargo:rerun-if-env-changed=MIRT_LIB_DIR");
}
