"""Builds the in-tree native libraries:
  csrc/libmor_hip.so    — HIP kernels + C ABI (hipcc, gfx950 only)
  csrc/libmor_synth.so  — synthetic LiDAR generator (gcc)
The .so files stay in-tree (git-ignored) so they travel with the gpurun snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HIP_SOURCES = ["mor_kernels.hip", "mor_engine.cpp"]
HIP_HEADERS = ["mor_device.h", os.path.join("..", "..", "include", "mor_hip.h")]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False):
    out = os.path.join(CSRC, "libmor_hip.so")
    deps = [os.path.join(CSRC, f) for f in HIP_SOURCES + HIP_HEADERS]
    if not force and not _newer(out, deps):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
           "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-x", "hip"] + [os.path.join(CSRC, f) for f in HIP_SOURCES] + ["-o", out]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


def build_hip_smalllist(force=False):
    """Test-only variant of the library with two-entry per-wave candidate lists in k_cg_slab, so that small scenes
    exercise the global overflow list (tests/test_gpu_parity.py::test_deferred_pair_overflow_list)."""
    out = os.path.join(CSRC, "libmor_hip_smalllist.so")
    deps = [os.path.join(CSRC, f) for f in HIP_SOURCES + HIP_HEADERS]
    if not force and not _newer(out, deps):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-DCGS_LISTW=16",
                           "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-x", "hip"] + [os.path.join(CSRC, f) for f in HIP_SOURCES] + ["-o", out])
    return out


def build_synth(force=False):
    out = os.path.join(CSRC, "libmor_synth.so")
    src = os.path.join(CSRC, "mor_synth.c")
    if not force and not _newer(out, [src]):
        return out
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", "-o", out, src, "-lm"])
    return out


def build_replay(force=False):
    """The drop-in class (include/MOR/MovingObjectRemoval.h over the C ABI) + the ROS-free replay driver."""
    out = os.path.join(CSRC, "mor_replay")
    srcs = [os.path.join(CSRC, "mor_adapter.cpp"), os.path.join(CSRC, "mor_replay.cpp")]
    inc = os.path.join(HERE, "..", "include")
    deps = srcs + [os.path.join(inc, "MOR", "MovingObjectRemoval.h"), os.path.join(inc, "MOR", "IncludeAll.h"), os.path.join(inc, "mor_hip.h"), os.path.join(CSRC, "libmor_hip.so")]
    if not force and not _newer(out, deps):
        return out
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-I", inc, "-o", out] + srcs + ["-L", CSRC, "-lmor_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"])
    return out


def build_all(force=False, verbose=False):
    return build_hip(force, verbose), build_synth(force), build_replay(force), build_hip_smalllist(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
