"""Development check of the grid stage's intermediate arrays against numpy (run through gpurun):
   python exp/check_grid.py [sensor] [seed] [frames]
Verifies for one stream: distinct keys ascending = np.unique of the point keys, cell ranges = counts, every point placed in
its cell's range of `sorted` exactly once, row table, smallest index / boxes per cell, slab boundaries, cell hash, and the
components of the merged slab forests against scipy on the oracle's edge definition (cluster labels are compared by the tests)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicslamtool_amd import kitti_params, synth
from dynamicslamtool_amd.engine import MorBatch

sensor = sys.argv[1] if len(sys.argv) > 1 else "hdl64"
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2001
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 2
p = kitti_params(1)
n = synth.n_points(sensor)
b = MorBatch(p, 1, n)
bad = 0


def check(name, cond):
    global bad
    if not cond:
        bad += 1
    print(("ok   " if cond else "FAIL ") + name)


for f in range(frames):
    x, pose = synth.frame(seed, sensor, f)
    b.push([x], pose[None, :])
    b.filter(to_host=False)
    cfg = b.debug_config()
    c = b.counts(0)
    M = int(c.n_cloud)
    nocc = b.stage_counts(0)["n_occ"]
    print("frame", f, cfg, "M", M, "n_occ", nocc)
    pkey = b.debug_read("pkey", 0, np.int32, M)
    ckey = b.debug_read("ckey", 0, np.int32, nocc)
    cstart = b.debug_read("cstart", 0, np.int32, nocc + 1)
    u, cnt = np.unique(pkey, return_counts=True)
    check("n_occ", nocc == len(u))
    check("ckey == unique(pkey)", len(u) == len(ckey) and np.array_equal(u, ckey))
    check("cstart == cumsum(counts)", len(u) == nocc and np.array_equal(cstart, np.concatenate([[0], np.cumsum(cnt)])))
    pcell = b.debug_read("pcell", 0, np.int32, M)
    check("pcell", np.array_equal(ckey[np.clip(pcell, 0, max(nocc - 1, 0))], pkey) and pcell.min(initial=0) >= 0 and pcell.max(initial=0) < max(nocc, 1))
    srt = b.debug_read("sorted", 0, np.float32, 4 * M).reshape(-1, 4)
    sidx = srt[:, 3].copy().view(np.int32)
    check("sorted is a permutation of the cloud", np.array_equal(np.sort(sidx), np.arange(M)))
    cloud = b.debug_read("cloud", 0, np.float32, 4 * M).reshape(-1, 4)
    check("sorted xyz", np.array_equal(srt[:, :3].view(np.uint32), cloud[np.clip(sidx, 0, M - 1), :3].view(np.uint32)))
    cell_of_pos = np.repeat(np.arange(nocc), np.diff(cstart)) if nocc else np.zeros(0, int)
    check("points lie in their cell's range", len(cell_of_pos) == M and np.array_equal(pcell[np.clip(sidx, 0, M - 1)], cell_of_pos))
    nx, ny, nz, nrows = cfg["nx"], cfg["ny"], cfg["nz"], cfg["nrows"]
    rs = b.debug_read("row_start", 0, np.int32, nrows + 1)
    check("row table", np.array_equal(rs, np.searchsorted(ckey, np.arange(nrows + 1, dtype=np.int64) * nx)))
    cmin = b.debug_read("cmin", 0, np.int32, nocc)
    order = np.argsort(cell_of_pos, kind="stable")
    want_min = np.minimum.reduceat(sidx, cstart[:-1]) if nocc else np.zeros(0, np.int32)
    check("cmin", np.array_equal(cmin, want_min))
    cm = b.debug_read("cmeta", 0, np.float32, 8 * nocc).reshape(-1, 8)
    if nocc:
        lo = np.stack([np.minimum.reduceat(srt[:, a], cstart[:-1]) for a in range(3)], 1)
        hi = np.stack([np.maximum.reduceat(srt[:, a], cstart[:-1]) for a in range(3)], 1)
        check("cell boxes", np.array_equal(cm[:, 0:3], lo) and np.array_equal(cm[:, 4:7], hi))
    P = cfg["P"]
    sy, sc, se = (b.debug_read(k, 0, np.int32, 33)[: P + 1] for k in ("slab_y", "slab_c", "slab_e"))
    print("   slabs y", sy.tolist(), "c", sc.tolist(), "e", se.tolist())
    ok = sy[0] == 0 and sy[P] == ny and sc[0] == 0 and sc[P] == nocc and all(sy[j + 1] >= sy[j] for j in range(P))
    ok = ok and all(sc[j] == rs[sy[j] * nz] for j in range(P + 1)) and all(se[j] == rs[min(sy[j + 1] + 2, ny) * nz] for j in range(P)) and all(sy[j + 1] - sy[j] >= 2 or sy[j + 1] == ny for j in range(P))
    check("slab boundaries", bool(ok))
    if cfg["cg_mode"] == 1 and nocc:
        # components of the merged forests == components the final labels imply (per cell: cluster id or unkept component)
        la, lb = b.debug_read("lroot_a", 0, np.int32, nocc), b.debug_read("lroot_b", 0, np.int32, nocc)
        ok = True
        for j in range(P):
            if sc[j + 1] > sc[j]:
                ok = ok and la[sc[j]:sc[j + 1]].min() >= sc[j] and la[sc[j]:sc[j + 1]].max() < se[j]
                if se[j] > sc[j + 1]:
                    ok = ok and lb[sc[j + 1]:se[j]].min() >= sc[j] and lb[sc[j + 1]:se[j]].max() < se[j]
        check("local roots lie in their slab's range", bool(ok))
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
