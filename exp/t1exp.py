"""Attribution of k_score_fast (tier 1 of the method-1 scores): library built with -DMOR_EXP_T1CUT as exp/libmor_t1exp.so (same hipcc line as
dynamicslamtool_amd/build.py), MOR_EXP_T1 = cut point (mor_kernels.hip).  Round 4, alone, B = 64 × 120 000 points: empty launch 6.7 µs, + cell index in LDS 8.8,
+ the query's point and cluster id 13.6, + cell lookup and the loads of the cell / cluster records 21.4, + sampled scan and classification 56.3, + worklist pushes
and counts 66.  Results are wrong for cut points > 0; only durations are read."""
import os, subprocess, sys
code = r'''
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from dynamicslamtool_amd import engine, kitti_params, synth
B, npts = 64, 120000
p = kitti_params(1)
seeds = [2000 + s for s in range(B)]
buf = engine.DeviceBuffer(3 * B * npts * 16); poses = []
for f in range(3):
    xs, ps = synth.batch(seeds, [f] * B); buf.upload(xs, f * B * npts * 16); poses.append(ps)
b = engine.MorBatch(p, B, npts)
views = [b.make_views([(buf.ptr + (f * B + s) * npts * 16, npts) for s in range(B)]) for f in range(3)]
def run(i):
    try:
        b.push_views(views[i % 3], poses[i % 3]); b.filter_device()
    except Exception as e:
        pass
for i in range(3): run(i)
b.kernel_timing_enable(True); b.kernel_timing(reset=True)
for i in range(3, 12): run(i)
kt = b.kernel_timing(reset=True)
print("RESULT", {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in kt.items() if k in ("score_fast", "score_nb", "score_pde", "split", "out")})
'''
for v in (0, 3, 4, 5):
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MOR_EXP_T1=str(v), MOR_HIP_LIB="/root/repo/exp/libmor_t1exp.so"), capture_output=True, text=True, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    print("variant %d:" % v, line[0] if line else "FAILED " + r.stderr[-300:])
