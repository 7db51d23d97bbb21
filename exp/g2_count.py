import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicslamtool_amd import engine, kitti_params, synth
B = 16
p = kitti_params(1); p.ground_method = 1
b = engine.MorBatch(p, B, 120000)
for f in range(3):
    xs, ps = synth.batch([2000 + s for s in range(B)], [f] * B); b.push(list(xs), ps); b.filter(to_host=False)
for s in range(B):
    n = int(b.debug_read("g2_nbig", s, np.int32, 1)[0]); big = b.debug_read("g2_big", s, np.int32, max(n, 1))[:n]
    print("stream %d: voxels %d, queued %d, left for the big kernel %d" % (s, b.stage_counts(s)["n_occ"], n, int((big >= 0).sum())))
