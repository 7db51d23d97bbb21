"""Parameter block + config-file grammar of the reference.

`MorParams` mirrors include/mor_hip.h:mor_params (the test checker uses the same
layout).  `parse_config` follows MovingObjectRemoval::setVariables
(/root/reference/src/MovingObjectRemoval.cpp:698-864): `key:value` lines, `#` comments, lines
shorter than 3 characters skipped, EVERY ':' stripped, no whitespace trimming, unknown key is an
error (the reference prints and exit(0)s; here: ValueError).
"""
import ctypes as C


class MorParams(C.Structure):
    _fields_ = [
        ("gp_limit", C.c_float), ("gp_leaf", C.c_float), ("bin_gap", C.c_float),
        ("min_cluster_size", C.c_int64), ("max_cluster_size", C.c_int64),
        ("volume_constraint", C.c_float), ("pde_lb", C.c_float), ("pde_ub", C.c_float),
        ("leave_off_distance", C.c_float), ("catch_up_distance", C.c_float),
        ("trim_x", C.c_float), ("trim_y", C.c_float), ("trim_z", C.c_float),
        ("ec_distance_threshold", C.c_float), ("pde_distance_threshold", C.c_float),
        ("method_choice", C.c_int32), ("opc_normalization_factor", C.c_int32),
        ("ground_method", C.c_int32), ("opc_resolution", C.c_float), ("volume_abs_int", C.c_int32),
        ("opc_anchor", C.c_int32),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_FLOAT_KEYS = {"gp_limit", "gp_leaf", "bin_gap", "volume_constraint", "pde_lb", "pde_ub",
               "leave_off_distance", "catch_up_distance", "trim_x", "trim_y", "trim_z",
               "ec_distance_threshold", "pde_distance_threshold"}
_LONG_KEYS = {"min_cluster_size", "max_cluster_size"}
_STRING_KEYS = {"output_topic", "debug_topic", "marker_topic", "input_pointcloud_topic",
                "input_odometry_topic", "output_fid", "debug_fid"}


def parse_config_text(text):
    """Returns (MorParams, dict of the string keys).  Grammar: MovingObjectRemoval.cpp:709-733."""
    p = MorParams()
    p.opc_resolution = 0.1  # hard-coded at the call site :575
    p.ground_method = 0     # :526 active, :527 commented out
    strings = {}
    for line in text.split("\n"):
        if len(line) < 3 or line[0] == "#":
            continue
        key, val, seen = "", "", False
        for ch in line:
            if ch == ":":
                seen = True
                continue
            if seen:
                val += ch
            else:
                key += ch
        if key in _FLOAT_KEYS:
            setattr(p, key, float(val))
        elif key in _LONG_KEYS:
            setattr(p, key, int(val))
        elif key == "method_choice":
            p.method_choice = int(val)
        elif key == "opc_normalization_factor":
            p.opc_normalization_factor = int(float(val))  # stof into an int, :843
        elif key == "ground_method":  # extension key: 0 crop, 1 voxel covariance
            p.ground_method = int(val)
        elif key == "volume_abs_int":  # extension key: 1 = the unqualified abs() of :277 truncates to int first (old libstdc++)
            p.volume_abs_int = int(val)
        elif key == "opc_anchor":  # extension key: method-2 voxel lattice anchored at p0 − res (0, default) or p0 − res/2 (1)
            p.opc_anchor = int(val)
        elif key in _STRING_KEYS:
            strings[key] = val
        else:
            raise ValueError("Invalid parameter found in config file: %r" % key)
    return p, strings


def parse_config(path):
    with open(path, "r") as f:
        return parse_config_text(f.read())


REF_DEFAULT_CONFIG = """#Detection algorithm choice
method_choice:2
output_topic:/output
debug_topic:/check
marker_topic:/bbox
input_pointcloud_topic:/velodyne_points
input_odometry_topic:/camera/odom/sample
output_fid:/filtered
debug_fid:/debug
ec_distance_threshold:0.11
min_cluster_size:200
max_cluster_size:35000
gp_leaf:0.1
bin_gap:10
gp_limit:-0.5
trim_x:3.0
trim_y:3.0
trim_z:5.0
pde_lb:0.005
pde_ub:0.5
pde_distance_threshold:0.15
opc_normalization_factor:20
volume_constraint:0.3
leave_off_distance:0.5
catch_up_distance:0.3
"""

# KITTI-scale profile proposed in SURVEY.md §8d (not from the reference)
KITTI_CONFIG = """method_choice:1
trim_x:50
trim_y:50
trim_z:2.5
gp_limit:-1.45
ec_distance_threshold:0.5
min_cluster_size:30
max_cluster_size:35000
pde_lb:0.005
pde_ub:0.5
pde_distance_threshold:0.15
volume_constraint:0.3
leave_off_distance:0.5
catch_up_distance:0.3
gp_leaf:0.2
bin_gap:10
opc_normalization_factor:20
"""


def ref_default_params():
    """Values of /root/reference/config/MOR_config.txt:1-39."""
    return parse_config_text(REF_DEFAULT_CONFIG)[0]


def kitti_params(method_choice=1):
    p = parse_config_text(KITTI_CONFIG)[0]
    p.method_choice = method_choice
    return p
