// build.rs — compile the HIP library for gfx950 and link it (uncompiled here: no Rust toolchain in the build image).
use std::{env, path::PathBuf, process::Command};

fn main() {
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let src = PathBuf::from(env::var("RT_MI355X_SRC").unwrap_or_else(|_| "raytracer_amd/csrc".into()));
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".into());
    let lib = out.join("librt_mi355x.so");
    let status = Command::new(&hipcc)
        .args(["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o"])
        .arg(&lib)
        .arg(src.join("rt_lib.hip"))
        .arg(src.join("rt_multi.hip"))
        .arg(src.join("rt_bvh.cpp"))
        .arg("-lpthread")
        .status()
        .expect("hipcc not found (set HIPCC)");
    assert!(status.success(), "hipcc failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=rt_mi355x");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", out.display());
    println!("cargo:rerun-if-changed={}", src.display());
    println!("cargo:rerun-if-env-changed=HIPCC");
}
