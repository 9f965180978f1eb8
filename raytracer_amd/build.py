"""Build recipe for the native libraries (gfx950 only).

    python -m raytracer_amd.build

* librt_mi355x.so        — HIP kernels + the C ABI of include/rt_abi.h   (hipcc --offload-arch=gfx950)
* libraytracer_host.so   — C++ host mirror of the reference API + C facade (links librt_mi355x)
Both are built in-tree (raytracer_amd/) so they travel to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIB_GPU = os.path.join(HERE, "librt_mi355x.so")
LIB_HOST = os.path.join(HERE, "libraytracer_host.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _sources(d):
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith((".hip", ".cpp", ".h", ".hpp"))]


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    abi = os.path.join(os.path.dirname(HERE), "include", "rt_abi.h")
    if force or _newer(LIB_GPU, _sources(CSRC) + [abi]):
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
               "-Wno-unused-function", "-o", LIB_GPU, os.path.join(CSRC, "rt_lib.hip"), os.path.join(CSRC, "rt_multi.hip"),
               os.path.join(CSRC, "rt_bvh.cpp"), "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    if force or _newer(LIB_HOST, _sources(HOST) + [abi, LIB_GPU]):
        # -ffp-contract=off: scene generation must round exactly like the reference (no fused a*b+c)
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-Wextra", "-o", LIB_HOST,
               os.path.join(HOST, "raytracer.cpp"), os.path.join(HOST, "host_capi.cpp"), "-L" + HERE, "-lrt_mi355x",
               "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    # C++ example driving the host layer the way the reference's main() does (examples/render_scene.cpp)
    example_src = os.path.join(os.path.dirname(HERE), "examples", "render_scene.cpp")
    example_bin = os.path.join(os.path.dirname(HERE), "examples", "render_scene")
    if os.path.exists(example_src) and (force or _newer(example_bin, [example_src, LIB_HOST, os.path.join(HOST, "raytracer.hpp")])):
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-I" + os.path.join(os.path.dirname(HERE), "include"),
               example_src, "-L" + HERE, "-lraytracer_host", "-lrt_mi355x", "-Wl,-rpath,$ORIGIN/../raytracer_amd",
               "-o", example_bin]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB_GPU, LIB_HOST


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
