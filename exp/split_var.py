"""Timing attribution of k_split: library built with -DMOR_EXP_SPLITVAR as exp/libmor_splitvar.so (same hipcc line as dynamicslamtool_amd/build.py);
MOR_SPLIT_VARIANT bit 0 = no look-back wait (prefixes wrong), bit 1 = no ground stores, bit 3 (8) = pure read of the tiles.  Results are wrong for
variants > 0; only the kernel's duration (synchronous steps, alone) is read.  Round 4, B = 64 × 120 000 points: full 81–98 µs, no wait + no ground
stores 62–74 µs, pure read 44 µs (123 MB: 2.8 TB/s)."""
import os, subprocess, sys
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
def run(i):
    try:
        b.push_views(views[i %% 2], poses[i %% 2]); b.filter_device()
    except Exception:
        pass
for i in range(3): run(i)
b.kernel_timing_enable(True); b.kernel_timing(reset=True)
for i in range(10): run(i)
kt = b.kernel_timing(reset=True)
print("RESULT", 1e3 * kt["split"][0] / kt["split"][1])
''' % ROOT
for v in (0, 8, 3):
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MOR_SPLIT_VARIANT=str(v), MOR_HIP_LIB=os.path.join(ROOT, "exp", "libmor_splitvar.so")), capture_output=True, text=True, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    print("variant %d: k_split alone %s us" % (v, line[0].split()[1] if line else "FAILED " + r.stderr[-300:]))
