"""Builds the in-tree native libraries:
  csrc/libmor_hip.so    — HIP kernels + C ABI (hipcc, gfx950 only)
  csrc/libmor_synth.so  — synthetic LiDAR generator (gcc)
  csrc/mor_replay       — the drop-in class + the ROS-free replay driver (g++); csrc/mor_replay_novis: the same without the VISUALIZE side effect
The .so files stay in-tree (git-ignored) so they travel with the gpurun snapshot.

An artefact is rebuilt when the hash of (its sources, its headers, its compiler flags) differs from the hash the artefact itself carries —
every build embeds `MOR_SRC_HASH=<sha256 prefix>` as a string in the binary (`built_hash` reads it back), so neither a fresh checkout's
mtimes nor a stale library that travelled with the tree can pass for a build of the present sources.  `stale()` is what
`dynamicslamtool_amd.engine.lib()` asks before loading: a library built from other sources fails loudly instead of being measured."""
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.normpath(os.path.join(HERE, "..", "include"))
HIP_SOURCES = ["mor_kernels.hip", "mor_engine.cpp"]
HIP_HEADERS = ["mor_device.h", os.path.join(INC, "mor_hip.h"), "kernels_common.h", "kernels_split.h", "kernels_grid.h", "kernels_cellgraph.h", "kernels_radix.h", "kernels_clusters.h",
               "kernels_scores.h", "kernels_ground_voxel.h", "kernels_track_out.h"]   # the kernels by stage: #included by mor_kernels.hip (one translation unit)
HIP_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
             "-Wall", "-Wno-unused-function", "-Wno-unused-value"]
SYNTH_FLAGS = ["-O2", "-fPIC", "-shared", "-fopenmp"]
REPLAY_FLAGS = ["-O2", "-std=c++17", "-Wall"]
_MARK = b"MOR_SRC_HASH="


def _path(f):
    return f if os.path.isabs(f) else os.path.join(CSRC, f)


def source_hash(files, flags):
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(_path(f), "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    h.update(" ".join(flags).encode())
    return h.hexdigest()[:24]


def built_hash(artefact):
    """The hash a built artefact carries (None: no artefact, or one from before hashes were embedded)."""
    try:
        with open(artefact, "rb") as fh:
            blob = fh.read()
    except OSError:
        return None
    m = re.search(re.escape(_MARK) + rb"([0-9a-f]{24})", blob)
    return m.group(1).decode() if m else None


def _define(h):
    return '-DMOR_SRC_HASH_STR="%s%s"' % (_MARK.decode(), h)


def _targets():
    hip_files = HIP_SOURCES + HIP_HEADERS
    adapter = ["mor_adapter.cpp", "mor_replay.cpp", os.path.join(INC, "MOR", "MovingObjectRemoval.h"), os.path.join(INC, "MOR", "IncludeAll.h"),
               os.path.join(INC, "MOR", "shim", "ros_pcl_types.h"), os.path.join(INC, "mor_hip.h")]
    return {
        "hip": (os.path.join(CSRC, "libmor_hip.so"), hip_files, HIP_FLAGS),
        "hip_smalllist": (os.path.join(CSRC, "libmor_hip_smalllist.so"), hip_files, HIP_FLAGS + ["-DCGS_LISTW=16"]),
        # lab build (exp/shadow.sh, exp/shadow_g2.sh; never loaded by the product, never built by build_all): the product's flags + the duplicated-launch hook,
        # so that shadow prices are measured on a library whose flags cannot drift from the product build's
        "hip_experiments": (os.path.join(HERE, "..", "exp", "libmor_exp.so"), hip_files, HIP_FLAGS + ["-DMOR_EXPERIMENTS"]),
        "synth": (os.path.join(CSRC, "libmor_synth.so"), ["mor_synth.c"], SYNTH_FLAGS),
        "replay": (os.path.join(CSRC, "mor_replay"), adapter, REPLAY_FLAGS),
        # the class without the reference's VISUALIZE side effect (IncludeAll.h:32 — a debug aid that is on by default there): what bench.py times as `class_latency_ms`
        "replay_novis": (os.path.join(CSRC, "mor_replay_novis"), adapter, REPLAY_FLAGS + ["-DMOR_NO_VISUALIZE"]),
    }


def stale(name="hip"):
    """True when the artefact is missing or was built from other sources / flags than the ones in the tree."""
    out, files, flags = _targets()[name]
    return built_hash(out) != source_hash(files, flags)


def _build(name, force, verbose, make_cmd):
    out, files, flags = _targets()[name]
    want = source_hash(files, flags)
    if not force and built_hash(out) == want:
        if verbose:
            print("%s: up to date (sources + flags hash %s)" % (os.path.basename(out), want), file=sys.stderr)
        return out
    cmd = make_cmd(out, flags + [_define(want)])
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    got = built_hash(out)
    if got != want:
        raise RuntimeError("%s does not carry the hash of its sources after the build (%r != %r)" % (out, got, want))
    return out


def _hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def build_hip(force=False, verbose=False):
    return _build("hip", force, verbose, lambda out, fl: [_hipcc()] + fl + ["-x", "hip"] + [os.path.join(CSRC, f) for f in HIP_SOURCES] + ["-o", out])


def build_hip_smalllist(force=False, verbose=False):
    """Test-only variant of the library with two-entry per-wave candidate lists in k_cg_slab, so that small scenes
    exercise the global overflow list (tests/test_gpu_parity.py::test_deferred_pair_overflow_list)."""
    return _build("hip_smalllist", force, verbose, lambda out, fl: [_hipcc()] + fl + ["-x", "hip"] + [os.path.join(CSRC, f) for f in HIP_SOURCES] + ["-o", out])


def build_hip_experiments(force=False, verbose=False):
    """exp/libmor_exp.so = HIP_FLAGS + -DMOR_EXPERIMENTS (MOR_EXP_DUP=<kernel id> launches a kernel twice: exp/shadow.sh)."""
    return _build("hip_experiments", force, verbose, lambda out, fl: [_hipcc()] + fl + ["-x", "hip"] + [os.path.join(CSRC, f) for f in HIP_SOURCES] + ["-o", out])


def build_synth(force=False, verbose=False):
    return _build("synth", force, verbose, lambda out, fl: ["gcc"] + fl + ["-o", out, os.path.join(CSRC, "mor_synth.c"), "-lm"])


def build_replay(force=False, verbose=False):
    """The drop-in class (include/MOR/MovingObjectRemoval.h over the C ABI) + the ROS-free replay driver."""
    srcs = [os.path.join(CSRC, "mor_adapter.cpp"), os.path.join(CSRC, "mor_replay.cpp")]
    return _build("replay", force, verbose, lambda out, fl: ["g++"] + fl + ["-I", INC, "-o", out] + srcs + ["-L", CSRC, "-lmor_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"])


def build_replay_novis(force=False, verbose=False):
    srcs = [os.path.join(CSRC, "mor_adapter.cpp"), os.path.join(CSRC, "mor_replay.cpp")]
    return _build("replay_novis", force, verbose, lambda out, fl: ["g++"] + fl + ["-I", INC, "-o", out] + srcs + ["-L", CSRC, "-lmor_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"])


def build_all(force=False, verbose=False):
    return build_hip(force, verbose), build_synth(force, verbose), build_replay(force, verbose), build_replay_novis(force, verbose), build_hip_smalllist(force, verbose)


if __name__ == "__main__":
    if "--experiments" in sys.argv:
        build_hip_experiments(force="--force" in sys.argv, verbose=True)
    else:
        build_all(force="--force" in sys.argv, verbose=True)
