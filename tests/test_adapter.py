"""The drop-in C++ class (include/MOR/MovingObjectRemoval.h + csrc/mor_adapter.cpp) driven by the
ROS-free replay driver (csrc/mor_replay.cpp), the counterpart of the reference's
src/external_sync_test.cpp callback."""
import os
import subprocess

import numpy as np
import pytest

from dynamicslamtool_amd.params import REF_DEFAULT_CONFIG
from scenes import scene_params, small_stream

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPLAY = os.path.join(ROOT, "dynamicslamtool_amd", "csrc", "mor_replay")


REF_CONFIG_FILE = os.path.join(ROOT, "tests", "golden", "MOR_config_reference.txt")   # the reference's config/MOR_config.txt, byte for byte


def _config_text(min_cluster_size=25, method=2):
    """The reference's REAL config file — comment blocks and blank lines intact — with two values changed for the small test scenes."""
    text = open(REF_CONFIG_FILE).read()
    assert text.count("\n\n") >= 6 and text.count("#") == 8
    return text.replace("min_cluster_size:200", "min_cluster_size:%d" % min_cluster_size).replace("method_choice:2", "method_choice:%d" % method)


def test_reference_config_file_through_the_class_constructor(tmp_path):
    """setVariables of the drop-in class on the reference's own config file, unmodified: every one of its 24 keys is echoed as the
    reference echoes it (.cpp:735-861), comment lines and blank lines are skipped, nothing is rejected.  (Without a GPU the
    constructor then fails in mor_create — after the parse; with one it goes on.)"""
    assert os.path.exists(REPLAY)
    r = subprocess.run([REPLAY, REF_CONFIG_FILE, "/dev/null", str(tmp_path), "/dev/null"], capture_output=True, text=True)
    assert "Invalid parameter" not in r.stdout and "Couldnt open" not in r.stdout
    want = [l for l in open(REF_CONFIG_FILE).read().split("\n") if len(l) >= 3 and l[0] != "#"]
    assert len(want) == 24
    echoed = r.stdout.split("\n")
    for l in want:
        key, val = l.split(":")
        assert any(e.startswith(key + ":") and (e == l or abs(float(e.split(":")[1]) - float(val)) < 1e-6) for e in echoed), (l, r.stdout)
    assert REF_DEFAULT_CONFIG.split("\n")[1] == "method_choice:2"


def test_config_errors_follow_the_reference(tmp_path):
    """Unknown key / unreadable file: message on stdout and exit(0), before any GPU work
    (reference .cpp:703-707, :856-860)."""
    assert os.path.exists(REPLAY), "build first: python -m dynamicslamtool_amd.build"
    bad = tmp_path / "bad.txt"
    bad.write_text("trim_x:3.0\nnot_a_key:1\n")
    r = subprocess.run([REPLAY, str(bad), "/dev/null", str(tmp_path), "/dev/null"], capture_output=True, text=True)
    assert r.returncode == 0 and "Invalid parameter found in config file" in r.stdout and "trim_x:3" in r.stdout
    r = subprocess.run([REPLAY, str(tmp_path / "missing.txt"), "/dev/null", str(tmp_path), "/dev/null"], capture_output=True, text=True)
    assert r.returncode == 0 and "Couldnt open the file" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("io", ["default", "page-locked"])
@pytest.mark.parametrize("method", [1, 2])
def test_replay_matches_oracle(tmp_path, method, io):
    """io = page-locked: the adapter's opt-in forms — the blob through a page-locked bounce buffer (method 1: read by the split kernel over PCIe; method 2: by DMA), the
    PointXYZI records written by the device straight into `output.data` — must leave the caller with the same bytes as the default (pageable in, host-side expansion)."""
    from oracle.oracle import Oracle
    env = dict(os.environ)
    if io == "page-locked":
        env.update(MOR_CLASS_INPUT="zerocopy" if method == 1 else "bounce", MOR_CLASS_OUTPUT="direct")
    cfg = tmp_path / "MOR_config.txt"
    cfg.write_text(_config_text(method=method))
    frames = small_stream(1, n_frames=8, with_nan=True)
    files = []
    with open(tmp_path / "poses.txt", "w") as pf:
        for i, (pts, pose) in enumerate(frames):
            fn = tmp_path / ("cloud_%04d.bin" % i)
            pts.astype(np.float32).tofile(fn)
            files.append(str(fn))
            pf.write(" ".join(repr(float(v)) for v in pose) + "\n")
    r = subprocess.run([REPLAY, str(cfg), str(tmp_path / "poses.txt"), str(tmp_path)] + files, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "method_choice:%d" % method in r.stdout   # the echo of setVariables
    p_ = scene_params(method_choice=method)
    o = Oracle(p_, 4, 3)
    removed = 0
    for i, (pts, pose) in enumerate(frames):
        o.push(pts, pose)
        raw = pts[np.isfinite(pts[:, :3]).all(1) & (np.abs(pts[:, 0]) <= p_.trim_x) & (np.abs(pts[:, 1]) <= p_.trim_y)]
        o_cloud = raw[(raw[:, 2] >= p_.gp_limit) & (raw[:, 2] <= p_.trim_z)]   # `cloud` (.cpp:85): indices of cluster_indices refer to it
        want = o.filter()
        got = np.fromfile(tmp_path / ("filtered_%04d.bin" % i), np.float32).reshape(-1, 4)
        assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), "frame %d" % i
        removed = max(removed, int(o.counts().n_trim) - len(want))
        # debug bounding boxes (mark_cluster, reference .cpp:7-58): one per cluster, centred on its centroid
        mk = np.loadtxt(tmp_path / ("markers_%04d.txt" % i), ndmin=2)
        cen = o.centroids()
        assert len(mk) == len(cen), "frame %d" % i
        if len(cen):
            pos, scale = o.markers()   # mark_cluster (.cpp:7-58): float-accumulated centroid, extent with 0 → 0.1
            assert np.allclose(mk[:, 1:4], pos, rtol=0, atol=2e-6) and np.allclose(mk[:, 4:7], scale, rtol=0, atol=2e-6), "frame %d" % i
            assert np.array_equal(mk[:, 7].astype(bool), o.detection().astype(bool)), "frame %d" % i
        # VISUALIZE (on by default, as in the reference's IncludeAll.h:32): from the second push on the caller's cloud and `output`
        # hold the clustered points of the new frame as 32-byte PointXYZI records (.cpp:553-558)
        pushed = np.fromfile(tmp_path / ("pushed_%04d.bin" % i), np.uint8)
        if i == 0:
            assert np.array_equal(pushed, pts.astype(np.float32).view(np.uint8).reshape(-1)), "first push leaves the caller's cloud alone (ca not initialised, .cpp:534)"
        else:
            off, idx = o.clusters()
            cc = o_cloud[idx]   # cluster_collection: the clustered points in cluster order
            rec = pushed.reshape(-1, 32)
            assert len(rec) == len(cc), "frame %d" % i
            got_xyz = rec[:, 0:12].copy().view(np.float32).reshape(-1, 3)
            got_i = rec[:, 16:20].copy().view(np.float32).reshape(-1)
            assert np.array_equal(got_xyz.view(np.uint32), cc[:, :3].view(np.uint32)) and np.array_equal(got_i.view(np.uint32), cc[:, 3].view(np.uint32)), "frame %d" % i
            assert ("pushed %d: caller cloud width %d point_step 32 output.width %d output.frame_id %s" % (i, len(cc), len(cc), "/debug")) in r.stdout, r.stdout
        # filterCloud: the incoming cloud's header travels through to out_cloud and `output` (.cpp:690-691), frame_id replaced (.cpp:692)
        assert ("frame %d:" % i) in r.stdout and ("frame_id /filtered, seq %d, stamp %d.000000000 (%d ns), cloud seq %d stamp %d" % (i, 100 + i, 1000000000 * (100 + i), i, 1000000 * (100 + i))) in r.stdout, r.stdout
    if method == 2:
        assert removed > 0


def test_adapter_compiles_against_ros_shaped_types(tmp_path):
    """INTEGRATION.md's build (`-DMOR_WITH_ROS_PCL`): mor_adapter.cpp must compile against headers shaped like the REAL ROS / PCL types —
    ros::Time with an explicit constructor (no assignment from double), allocator-templated messages, pcl::uint8_t vectors,
    pcl_conversions::fromPCL — not only against the in-repo shim.  Compile-only (tests/ros_stub/ are data carriers)."""
    src = os.path.join(ROOT, "dynamicslamtool_amd", "csrc", "mor_adapter.cpp")
    for extra in ([], ["-DMOR_NO_VISUALIZE"], ["-DINTERNAL_SYNC"], ["-DINTERNAL_SYNC", "-DMOR_NO_VISUALIZE"]):   # the reference's two compile-time flags (IncludeAll.h:32, :36)
        r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-c", "-DMOR_WITH_ROS_PCL", "-I", os.path.join(ROOT, "tests", "ros_stub"), "-I", os.path.join(ROOT, "include"),
                            src, "-o", str(tmp_path / "adapter.o")] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def _build_internal_sync_node(tmp_path):
    """The class with -DMOR_WITH_ROS_PCL -DINTERNAL_SYNC against tests/ros_stub + the test node that stands in for internal_sync_test.cpp."""
    exe = str(tmp_path / "internal_sync_node")
    csrc = os.path.join(ROOT, "dynamicslamtool_amd", "csrc")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-DMOR_WITH_ROS_PCL", "-DINTERNAL_SYNC", "-I", os.path.join(ROOT, "tests", "ros_stub"), "-I", os.path.join(ROOT, "include"),
                        "-o", exe, os.path.join(ROOT, "tests", "ros_stub", "internal_sync_node.cpp"), os.path.join(csrc, "mor_adapter.cpp"), "-L", csrc, "-lmor_hip",
                        "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_internal_sync_build_links_and_parses_the_reference_config(tmp_path):
    """SURVEY §8f-4, compile + link level (CPU): the constructor's advertise / subscribe / synchroniser block (.cpp:372-385) and
    movingCloudObjectSubscriber (.cpp:393-413) build against ROS-shaped headers, and the node gets as far as the reference's does without a
    device: the config echo, then mor_create's refusal (no CPU fallback)."""
    exe = _build_internal_sync_node(tmp_path)
    r = subprocess.run([exe, REF_CONFIG_FILE, "/dev/null", str(tmp_path), "/dev/null"], capture_output=True, text=True)
    assert "input_pointcloud_topic:/velodyne_points" in r.stdout and "input_odometry_topic:/camera/odom/sample" in r.stdout and "Invalid parameter" not in r.stdout


@pytest.mark.gpu
def test_internal_sync_callback_publishes_what_the_external_sync_calls_return(tmp_path):
    """SURVEY §8f-4 end to end without ROS: clouds and odometry published into the class's own subscriptions (in-process bus of tests/ros_stub) come
    out on output_topic exactly as pushRawCloudAndPose + filterCloud return them to an external-sync caller (mor_replay) and as the oracle says;
    the debug topic carries the clustered points from the second frame on (.cpp:553-558); the marker topic one CUBE per tracked centroid the
    filterCloud loop visited — ids 1, 2, …, the matched cluster's float centroid and extent, the reference's colour and lifetime (.cpp:7-58, :623, :641)."""
    from oracle.oracle import Oracle
    exe = _build_internal_sync_node(tmp_path)
    cfg = tmp_path / "MOR_config.txt"
    cfg.write_text(_config_text(method=2))
    frames = small_stream(1, n_frames=8, with_nan=True)
    files = []
    with open(tmp_path / "poses.txt", "w") as pf:
        for i, (pts, pose) in enumerate(frames):
            fn = tmp_path / ("cloud_%04d.bin" % i)
            pts.astype(np.float32).tofile(fn)
            files.append(str(fn))
            pf.write(" ".join(repr(float(v)) for v in pose) + "\n")
    ext, internal = tmp_path / "ext", tmp_path / "int"
    ext.mkdir()
    internal.mkdir()
    r1 = subprocess.run([REPLAY, str(cfg), str(tmp_path / "poses.txt"), str(ext)] + files, capture_output=True, text=True)
    assert r1.returncode == 0, r1.stdout + r1.stderr
    r2 = subprocess.run([exe, str(cfg), str(tmp_path / "poses.txt"), str(internal)] + files, capture_output=True, text=True)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    assert r2.stdout.count("-----------------------------------------------------") == 2 * len(frames)   # the callback's two rules per iteration (.cpp:398, :412)
    p_ = scene_params(method_choice=2)
    o = Oracle(p_, 4, 3)
    n_markers = 0
    for i, (pts, pose) in enumerate(frames):
        o.push(pts, pose)
        off, idx = o.clusters()
        raw = pts[np.isfinite(pts[:, :3]).all(1) & (np.abs(pts[:, 0]) <= p_.trim_x) & (np.abs(pts[:, 1]) <= p_.trim_y)]
        o_cloud = raw[(raw[:, 2] >= p_.gp_limit) & (raw[:, 2] <= p_.trim_z)]
        want = o.filter()
        got = np.fromfile(internal / ("filtered_%04d.bin" % i), np.float32).reshape(-1, 4)
        same = np.fromfile(ext / ("filtered_%04d.bin" % i), np.float32).reshape(-1, 4)
        assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)) and np.array_equal(got.view(np.uint32), same.view(np.uint32)), "frame %d" % i
        assert ("frame %d: published %d pts, frame_id /filtered, seq %d, stamp %d ns, debug clouds %d" % (i, len(want), i, 1000000000 * (100 + i), 1 if i else 0)) in r2.stdout, r2.stdout
        if i:
            dbg = np.fromfile(internal / ("debug_%04d.bin" % i), np.float32).reshape(-1, 4)
            assert np.array_equal(dbg.view(np.uint32), o_cloud[idx].astype(np.float32).view(np.uint32)), "frame %d" % i
        # markers: one per tracked centroid visited by the loop, on the cluster it was matched to
        mc = o.moving_clusters()
        rows = [l.split() for l in open(internal / ("moving_markers_%04d.txt" % i)).read().splitlines()]
        assert len(rows) == len(mc), "frame %d" % i
        if len(mc):
            pos, scale = o.markers()
            for j, row in enumerate(rows):
                assert int(row[0]) == j + 1 and row[12] == "bounding_box" and row[13] == "/debug" and int(row[14]) == 1 and int(row[15]) == 0
                assert np.allclose([float(v) for v in row[1:4]], pos[mc[j]], rtol=0, atol=2e-6) and np.allclose([float(v) for v in row[4:7]], scale[mc[j]], rtol=0, atol=2e-6)
                assert np.allclose([float(v) for v in row[7:12]], [0.8, 0.1, 0.4, 0.5, 2.0], rtol=0, atol=1e-6)
        n_markers += len(mc)
    assert n_markers > 0
