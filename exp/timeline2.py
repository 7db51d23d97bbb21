"""Untraced overlap analysis: HIP-event timestamps of every launch of a pipelined leg (mor_kernel_timeline_read)."""
import os, sys, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicslamtool_amd import engine, kitti_params, synth
B, npts, nf, steps = 64, 120000, 12, 40
p = kitti_params(1)
seeds = [2000 + s for s in range(B)]
buf = engine.DeviceBuffer(nf * B * npts * 16); poses = np.empty((nf, B, 7))
for f in range(nf):
    xs, ps = synth.batch(seeds, [f] * B); buf.upload(xs, f * B * npts * 16); poses[f] = ps
b = engine.MorBatch(p, B, npts)
views = [b.make_views([(buf.ptr + (f * B + s) * npts * 16, npts) for s in range(B)]) for f in range(nf)]
def fr(i):
    k = i % (2 * (nf - 1)); return k if k < nf else 2 * (nf - 1) - k
for i in range(6): b.push_views(views[fr(i)], poses[fr(i)]); b.filter_device()
b.kernel_timing_enable(True); b.kernel_timing(reset=True)
b.set_async(True)
for i in range(6, 6 + steps): b.push_views(views[fr(i)], poses[fr(i)]); b.filter_async()
b.wait(); b.set_async(False)
b.kernel_timing(reset=True)
tl = b.kernel_timeline()
first = "classify" if any(n == "classify" for n, _, _ in tl) else "split"
st = sorted(t0 for n, t0, t1 in tl if n == first)
T0, T1 = st[int(len(st) * 0.25)], st[int(len(st) * 0.85)]
nsteps = int(len(st) * 0.85) - int(len(st) * 0.25)
win = [(max(a, T0), min(c, T1), n) for n, a, c in tl if c > T0 and a < T1]
T = T1 - T0
print("window %.2f ms, %d steps → %.1f us per step (with per-kernel event timing on)" % (T, nsteps, 1e3 * T / nsteps))
ev = sorted([(a, 1) for a, c, n in win] + [(c, -1) for a, c, n in win])
conc = collections.Counter(); cur = 0; last = T0
for t, d in ev:
    conc[cur] += t - last; last = t; cur += d
print("concurrency: " + "  ".join("%d: %.1f%%" % (k, 100 * v / T) for k, v in sorted(conc.items())), " mean %.2f" % (sum(k * v for k, v in conc.items()) / T))
dur = collections.Counter()
for a, c, n in win: dur[n] += c - a
print("kernel us per step: " + ", ".join("%s %.0f" % (n, 1e3 * v / nsteps) for n, v in dur.most_common(12)))
# one step in detail
k0 = st[int(len(st) * 0.5)]
print("--- launches starting within 900 us of one classify launch")
for n, a, c in sorted(tl, key=lambda x: x[1]):
    if k0 - 0.05 <= a < k0 + 0.9: print("  %8.1f %8.1f  %-14s %6.1f" % (1e3 * (a - k0), 1e3 * (c - k0), n, 1e3 * (c - a)))
