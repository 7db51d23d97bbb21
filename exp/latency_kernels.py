"""Where the latency of ONE stream goes (B = 1, synchronous push + filter of a device-resident 120 000-point cloud): per-kernel HIP-event times."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicslamtool_amd import engine, kitti_params, synth
p = kitti_params(1)
b = engine.MorBatch(p, 1, 120000)
frames = [synth.frame(2005, "hdl64", f) for f in range(14)]
buf = engine.DeviceBuffer(120000 * 16)
ts = []
for f, (x, pose) in enumerate(frames):
    buf.upload(x)
    if f == 4:
        b.kernel_timing_enable(True); b.kernel_timing(reset=True)
    t0 = time.perf_counter()
    b.push([(buf, 120000)], pose[None, :]); b.filter(to_host=False)
    ts.append(time.perf_counter() - t0)
kt = b.kernel_timing(reset=True)
n = len(frames) - 4
print("device-resident push + filter: median %.3f ms (with event timing on: %.3f)" % (1e3 * np.median(ts[1:4]), 1e3 * np.median(ts[4:])))
tot = 0
for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0]):
    if v[1]:
        print("  %-14s %6.1f us" % (k, 1e3 * v[0] / n)); tot += 1e3 * v[0] / n
print("  sum %.1f us" % tot)
