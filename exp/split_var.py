"""Timing attribution of k_split (results are wrong for variants ≥ 2; only the kernel's duration is read)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from dynamicslamtool_amd import engine, kitti_params, synth
B, npts = 64, 120000
p = kitti_params(1)
seeds = [2000 + s for s in range(B)]
buf = engine.DeviceBuffer(2 * B * npts * 16); poses = []
for f in range(2):
    xs, ps = synth.batch(seeds, [f] * B); buf.upload(xs, f * B * npts * 16); poses.append(ps)
b = engine.MorBatch(p, B, npts)
views = [b.make_views([(buf.ptr + (f * B + s) * npts * 16, npts) for s in range(B)]) for f in range(2)]
for i in range(3): b.push_views(views[i %% 2], poses[i %% 2]); b.filter_device()
b.kernel_timing_enable(True); b.kernel_timing(reset=True)
for i in range(10): b.push_views(views[i %% 2], poses[i %% 2]); b.filter_device()
kt = b.kernel_timing(reset=True)
print("RESULT", 1e3 * kt["split"][0] / kt["split"][1])
''' % ROOT
for v in (0, 1, 2, 4, 6, 8, 16, 24):
    try:
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MOR_SPLIT_VARIANT=str(v)), capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        print("variant %2d: k_split alone %s us" % (v, line[0].split()[1] if line else "FAILED " + r.stderr[-300:]))
    except Exception as e:
        print("variant", v, "failed", e)
