"""What the lanes of the frame pipeline do between their kernels: from a rocprofv3 kernel trace (--kernel-trace --output-format csv) of an asynchronous leg, per hardware queue the share
of the steady-state window in which a kernel of that queue runs, the gaps in front of every kernel (by name: median / mean) and how many kernels run side by side over time.
usage: lane_idle.py <t_kernel_trace.csv> [first_frame last_frame]   (frames = launches of k_gridhash; default: the middle half of the longest burst)"""
import csv, sys, collections
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in rows]
ev.sort()
gh = [e for e in ev if e[3].startswith("k_gridhash")]
st = np.array([e[0] for e in gh])
# the asynchronous legs: launches of k_gridhash closer than 1 ms to each other; take the longest burst
d = np.diff(st); cut = np.where(d > 2e6)[0]
bounds = np.concatenate([[0], cut + 1, [len(st)]])
i = int(np.argmax(np.diff(bounds))); a, b = bounds[i], bounds[i + 1]
if len(sys.argv) > 3: a, b = a + int(sys.argv[2]), a + int(sys.argv[3])
else: a, b = a + (b - a) // 4, b - (b - a) // 4
t0, t1 = st[a], st[b - 1]
print("window: %d frames, %.1f us per frame" % (b - 1 - a, (t1 - t0) / 1e3 / (b - 1 - a)))
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
byq = collections.defaultdict(list)
for e in win: byq[e[2]].append(e)
gaps = collections.defaultdict(list)
for q, es in sorted(byq.items()):
    busy = sum(e[1] - e[0] for e in es)
    print("queue %d: %d kernels, busy %.1f %% of the window" % (q, len(es), 100.0 * busy / (t1 - t0)))
    for p, e in zip(es, es[1:]): gaps[e[3]].append((e[0] - p[1]) / 1e3)
print("gap in front of a kernel on its queue (us): median / mean / p90, and its own duration")
dur = collections.defaultdict(list)
for e in win: dur[e[3]].append((e[1] - e[0]) / 1e3)
for n in sorted(gaps, key=lambda n: -np.mean(gaps[n]) * len(gaps[n])):
    g = np.array(gaps[n]); print("  %-22s n %4d  gap %6.1f / %6.1f / %6.1f   runs %6.1f" % (n, len(g), np.median(g), g.mean(), np.percentile(g, 90), np.mean(dur[n])))
# concurrency histogram
pts = sorted([(e[0], 1) for e in win] + [(e[1], -1) for e in win])
cur, last, hist = 0, t0, collections.Counter()
for t, dlt in pts:
    hist[cur] += t - last; last = t; cur += dlt
tot = sum(hist.values())
print("kernels running side by side: " + "  ".join("%d: %.1f %%" % (k, 100.0 * v / tot) for k, v in sorted(hist.items())))
