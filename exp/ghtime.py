"""k_gridhash alone (synchronous steps), hdl64_b64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicslamtool_amd import engine, kitti_params, synth
B, npts = 64, 120000
p = kitti_params(1)
seeds = [2000 + s for s in range(B)]
buf = engine.DeviceBuffer(4 * B * npts * 16); poses = []
for f in range(4):
    xs, ps = synth.batch(seeds, [f] * B); buf.upload(xs, f * B * npts * 16); poses.append(ps)
b = engine.MorBatch(p, B, npts)
views = [b.make_views([(buf.ptr + (f * B + s) * npts * 16, npts) for s in range(B)]) for f in range(4)]
for i in range(4): b.push_views(views[i % 4], poses[i % 4]); b.filter_device()
b.kernel_timing_enable(True); b.kernel_timing(reset=True)
for i in range(12): b.push_views(views[i % 4], poses[i % 4]); b.filter_device()
kt = b.kernel_timing(reset=True)
print("gridhash alone %.1f us, gridplace %.1f, cg_slab %.1f" % tuple(1e3 * kt[k][0] / kt[k][1] for k in ("gridhash", "gridplace", "cg_slab")))
