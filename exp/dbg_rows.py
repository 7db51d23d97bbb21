import sys; sys.path.insert(0,'/root/repo')
from dynamicslamtool_amd import engine, kitti_params, synth
import numpy as np
b = engine.MorBatch(kitti_params(1), 64, 120000)
print(b.debug_config())
xs, ps = synth.batch([2000+s for s in range(64)], [0]*64)
b.push(list(xs), ps); b.filter(to_host=False)
n=[b.stage_counts(s)["n_occ"] for s in range(64)]
print(sorted(n)[-8:], np.mean(n))
