"""Full-BATCH digests (tests/golden/fullbatch_digests.json): every stream of the BASELINE 1-GPU configurations other than configs[1] — 64 × os128
(262 144 points), 32 × agg10 (1 000 000 points) — plus the 64 hdl64 streams with the voxel-covariance ground removal and with method 2, through the CPU oracle for the
first two frame pairs (3 frames).  The synthetic generator is deterministic and in-repo, so the fixture holds only counts and CRC-32 digests.
`record_of` is the ONE definition of a record, used by the generator (with the oracle), the CPU test (oracle again, a sample of the streams) and the GPU
tests (every stream of the batch through the C ABI).  TEST INFRASTRUCTURE."""
import json
import os
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "golden", "fullbatch_digests.json")
N_FRAMES = 3
CASES = {   # name: (sensor, BASELINE config index of the seeds, streams, method, ground method)
    "os128_b64": ("os128", 3, 64, 1, 0),
    "agg10_b32": ("agg10", 5, 32, 1, 0),
    "hdl64_b64_voxel_ground": ("hdl64", 2, 64, 1, 1),
    "hdl64_b64_method2": ("hdl64", 2, 64, 2, 0),
}
FIELDS = ("T", "M", "G", "K", "C", "pairs", "labels", "corr", "detection", "n_out", "out", "tracks")


def params_of(name):
    from dynamicslamtool_amd import kitti_params
    p = kitti_params(CASES[name][3])
    p.ground_method = CASES[name][4]
    return p


def seeds_of(name):
    _, cfg, B = CASES[name][:3]
    return [1000 * cfg + s for s in range(B)]


def crc(*arrays):
    return zlib.crc32(b"".join(np.ascontiguousarray(a).tobytes() for a in arrays)) & 0xFFFFFFFF


def record_before_filter(eng, s=None):
    a = (lambda n: getattr(eng, n)()) if s is None else (lambda n: getattr(eng, n)(s))
    c = a("counts")
    q, m, _, sc = a("correspondences")
    return [int(c.n_trim), int(c.n_cloud), int(c.n_ground), int(c.n_clusters), int(c.n_clustered), int(c.n_corr), crc(a("labels")),
            crc(np.asarray(q, np.int32), np.asarray(m, np.int32), np.asarray(sc, np.float64)), crc(a("detection"))]


def record_after_filter(eng, out, s=None):
    c = eng.counts() if s is None else eng.counts(s)
    return [int(len(out)), crc(out), int(c.n_tracks)]


def oracle_stream(job):
    """(case name, seed) → [record per frame] through the CPU oracle."""
    name, seed = job
    from dynamicslamtool_amd import synth
    from oracle.oracle import Oracle
    o, recs = Oracle(params_of(name), 4, 3), []
    for f in range(N_FRAMES):
        x, ps = synth.frame(seed, CASES[name][0], f)
        o.push(x, ps)
        r = record_before_filter(o)
        recs.append(r + record_after_filter(o, o.filter()))
    o.close()
    return recs


def load():
    return json.load(open(PATH))
