"""Idle time of a lane in front of every kernel of a frame (pipelined leg, HIP-event timestamps of every launch): frame i is the i-th
launch of each kernel, its kernels run one after the other on one lane, so start(next) − end(previous) is what the lane waited —
launch latency, plus, in front of k_clusters / k_track_push / k_split, the wait for another frame's event."""
import os, sys, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicslamtool_amd import engine, kitti_params, synth
B, npts, nf, steps = 64, 120000, 12, 60
p = kitti_params(1)
engine.bind_thread_to_device_node(0)
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
per = collections.defaultdict(list)
for n, a, c in tl: per[n].append((a, c))
for n in per: per[n].sort()
order = ["split", "gridcount", "gridhash", "gridplace", "cellboxes", "cg_slab", "clusters", "score_fast", "score_nb", "score_pde", "track_push", "track_filter", "out"]
nfr = min(len(per[n]) for n in order)
lo, hi = nfr // 4, nfr - 4
gaps = collections.defaultdict(list); durs = collections.defaultdict(list); frame_len = []
for i in range(lo, hi):
    prev_end = None
    for n in order:
        a, c = per[n][i]
        if prev_end is not None: gaps[n].append(1e3 * (a - prev_end))
        durs[n].append(1e3 * (c - a)); prev_end = c
    frame_len.append(1e3 * (per["out"][i][1] - per["split"][i][0]))
    if i >= 4: gaps["split"].append(1e3 * (per["split"][i][0] - per["out"][i - 4][1]))   # the lane's previous frame
period = 1e3 * (per["split"][hi - 1][0] - per["split"][lo][0]) / (hi - 1 - lo)
print("period %.1f us (event timing on); a frame takes %.0f us from its split to its output; kernels %.0f us, gaps %.0f us per frame" % (
    period, np.mean(frame_len), sum(np.mean(durs[n]) for n in order), sum(np.mean(gaps[n]) for n in order if n != "split")))
for n in order:
    print("  %-13s runs %6.1f us   lane idle in front of it: mean %6.1f  median %6.1f  p90 %6.1f" % (n, np.mean(durs[n]), np.mean(gaps[n]), np.median(gaps[n]), np.percentile(gaps[n], 90)))
