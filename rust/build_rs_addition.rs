// lines to add to the crate's build.rs (next to its bindgen call, build.rs:11-27)
// build.rs (next to the existing bindgen call, build.rs:11-27)
println!("cargo:rustc-link-search=native={}", std::env::var("GNSS_MI355X_LIB_DIR").unwrap());
println!("cargo:rustc-link-lib=dylib=gnss_mi355x");
