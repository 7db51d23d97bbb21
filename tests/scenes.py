"""Small synthetic scenes for parity tests (≤ a few thousand points, so the O(n²) brute force runs).

An indoor-robot-like scene matching the reference's default profile (trim ±3 m, r = 0.11): a noisy
floor, a few box-shaped objects sampled on their surfaces, some of them moving, some clutter, and
points outside the trim box.  Points are emitted in the sensor frame of a slowly moving ego pose.
"""
import numpy as np

from dynamicslamtool_amd import ref_default_params


def scene_params(method_choice=2, min_cluster_size=25):
    p = ref_default_params()
    p.method_choice = method_choice
    p.min_cluster_size = min_cluster_size
    return p


def _box_surface(rng, center, size, n):
    """n points on the surface of an axis-aligned box."""
    pts = (rng.random((n, 3)) - 0.5) * size
    face = rng.integers(0, 3, n)
    sign = rng.integers(0, 2, n) * 2 - 1
    pts[np.arange(n), face] = 0.5 * size[face] * sign
    return pts + center


def _yaw_quat(yaw):
    return np.array([0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2)])


def small_stream(seed, n_frames=7, n_objects=6, pts_per_object=220, n_floor=600, n_clutter=150, n_far=100, with_nan=True):
    """Returns a list of (xyzi float32 [N,4], pose float64 [7])."""
    rng = np.random.default_rng(seed)
    centers = np.column_stack([rng.uniform(-2.4, 2.4, n_objects), rng.uniform(-2.4, 2.4, n_objects), rng.uniform(-0.2, 0.6, n_objects)])
    sizes = rng.uniform(0.25, 0.6, (n_objects, 3))
    vel = np.zeros((n_objects, 3))
    n_moving = max(1, n_objects // 3)
    vel[:n_moving, :2] = rng.uniform(-0.09, 0.09, (n_moving, 2))
    # fixed local surface samples per object (rigid objects re-observed with small noise)
    local = [_box_surface(rng, np.zeros(3), sizes[k], pts_per_object + int(rng.integers(-40, 40))) for k in range(n_objects)]
    frames = []
    for f in range(n_frames):
        ego_xy = np.array([0.02 * f, 0.01 * f])
        ego_yaw = 0.01 * f
        world = []
        for k in range(n_objects):
            world.append(local[k] + centers[k] + vel[k] * f + rng.normal(0, 0.002, local[k].shape))
        floor = np.column_stack([rng.uniform(-3.5, 3.5, n_floor), rng.uniform(-3.5, 3.5, n_floor), rng.normal(-0.62, 0.01, n_floor)])
        clutter = np.column_stack([rng.uniform(-3, 3, n_clutter), rng.uniform(-3, 3, n_clutter), rng.uniform(-0.5, 2.0, n_clutter)])
        far = rng.uniform(-8, 8, (n_far, 3))
        w = np.concatenate(world + [floor, clutter, far], axis=0)
        # world → sensor frame
        c, s = np.cos(-ego_yaw), np.sin(-ego_yaw)
        d = w[:, :2] - ego_xy
        x = d[:, 0] * c - d[:, 1] * s
        y = d[:, 0] * s + d[:, 1] * c
        pts = np.column_stack([x, y, w[:, 2], rng.random(len(w))]).astype(np.float32)
        pts = pts[rng.permutation(len(pts))]
        if with_nan and len(pts) > 10:
            pts[3, 0] = np.nan
            pts[7, 2] = np.inf
        pose = np.concatenate([[ego_xy[0], ego_xy[1], 0.0], _yaw_quat(ego_yaw)])
        frames.append((pts, pose))
    return frames


def sweep_case(case):
    """Randomised parameter profile + two small streams of tests/test_gpu_parity.py::test_parameter_sweep_on_small_streams (shared with
    tests/golden/make_sweep_minimums.py, which records what the oracle produces for each case)."""
    rng = np.random.default_rng(1000 + case)
    p = scene_params(method_choice=int(rng.integers(1, 3)))
    p.ec_distance_threshold = float(rng.choice([0.06, 0.11, 0.18, 0.3]))
    p.min_cluster_size = int(rng.choice([5, 25, 60]))
    p.max_cluster_size = int(rng.choice([150, 400, 20000]))
    p.trim_x, p.trim_y = float(rng.choice([2.0, 3.0, 5.0])), float(rng.choice([2.0, 3.0, 5.0]))
    p.trim_z = float(rng.choice([0.8, 2.0]))
    p.gp_limit = float(rng.choice([-0.6, -0.55, -0.3]))
    p.volume_constraint = float(rng.choice([0.1, 0.3, 0.9]))
    lb, ub = [(0.0001, 0.003), (0.002, 0.02), (0.01, 0.5), (0.05, 0.01), (-1.0, 0.004), (0.0, 0.0002), (0.0005, 2.5)][case % 7]
    p.pde_lb, p.pde_ub = lb, ub
    p.pde_distance_threshold = float(rng.choice([0.05, 0.15, 0.5]))
    p.opc_normalization_factor = int(rng.choice([5, 15, 40]))
    p.leave_off_distance, p.catch_up_distance = float(rng.choice([0.05, 0.4])), float(rng.choice([0.1, 0.3]))
    n_bad, n_good = int(rng.integers(2, 6)), int(rng.integers(1, 5))
    # two draws left every component outside the size window (tolerance 0.06 with min 60; tolerance 0.3 with max 150 merges whole boxes):
    # a sweep case that forms no cluster exercises nothing behind the clustering, so those two get the neighbouring choice
    if case == 7:
        p.min_cluster_size = 5
    if case == 13:
        p.max_cluster_size = 400
    streams = [small_stream(100 + 3 * case + i, n_frames=8, n_objects=int(rng.integers(3, 9))) for i in range(2)]
    return p, n_bad, n_good, streams
