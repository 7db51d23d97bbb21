import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicslamtool_amd import engine, kitti_params, synth
B, npts, nf = 64, 120000, 20
p = kitti_params(1)
seeds = [2000 + s for s in range(B)]
frames = [synth.batch(seeds, [f] * B) for f in range(nf)]
a = engine.MorBatch(p, B, npts); b = engine.MorBatch(p, B, npts)
for f in range(nf):
    a.push(list(frames[f][0]), frames[f][1]); a.filter(to_host=False)
b.set_async(True)
for f in range(nf):
    b.push(list(frames[f][0]), frames[f][1]); b.filter_async()
b.wait()
bad = 0
for s in range(B):
    ca, cb = a.correspondences(s), b.correspondences(s)
    for x, y in zip(ca, cb):
        if not np.array_equal(np.asarray(x), np.asarray(y)): bad += 1
    oa, ob = a.output_device(s), b.output_device(s)
    if oa[1] != ob[1]: bad += 1
    ta, tb = a.tracks(s), b.tracks(s)
    for x, y in zip(ta, tb):
        if not np.array_equal(np.asarray(x), np.asarray(y)): bad += 1
print("mismatches", bad, "stage totals sync", {k: sum(a.stage_counts(s)[k] for s in range(B)) for k in ("n_tier1b", "n_defer")}, "async", {k: sum(b.stage_counts(s)[k] for s in range(B)) for k in ("n_tier1b", "n_defer")})
