// points rustc at libs2r.so (built by `python synth2_amd/build.py`); UNVERIFIED, see Cargo.toml
fn main() {
    let dir = std::env::var("S2R_LIB_DIR").unwrap_or_else(|_| "../../synth2_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=s2r");
}
