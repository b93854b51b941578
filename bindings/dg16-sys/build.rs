// libdg16.so is built by `make -C distributed-groth16_amd/csrc` of the library's repository (hipcc, gfx950 only).
//   DG16_LIB_DIR   directory that holds libdg16.so            (default: ../../distributed-groth16_amd)
// librccl / libamdhip64 are dependencies of libdg16.so itself (RCCL is bound with dlopen at run time): nothing else
// to link here.
use std::env;
use std::path::PathBuf;

fn main() {
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let dir = env::var("DG16_LIB_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|_| manifest.join("../../distributed-groth16_amd"));
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=dg16");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=DG16_LIB_DIR");
    println!("cargo:rerun-if-changed=build.rs");
    // handed to dependants as DEP_DG16_INCLUDE
    println!("cargo:include={}", manifest.join("../../include").display());
}
