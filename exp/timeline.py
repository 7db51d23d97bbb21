"""Overlap analysis of a rocprofv3 --kernel-trace CSV of the pipelined bench: average concurrency, busy fraction per queue,
idle gaps per queue, time during which a kernel runs alone.  Usage: python exp/timeline.py <t_kernel_trace.csv>"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), n))
rows.sort()
# steady state: middle 60 % of the k_classify / k_split launches
firsts = [a for a, b, q, n in rows if n in ("k_split", "k_classify")]
t0, t1 = firsts[int(len(firsts) * 0.2)], firsts[int(len(firsts) * 0.8)]
steps = int(len(firsts) * 0.8) - int(len(firsts) * 0.2)
win = [(max(a, t0), min(b, t1), q, n) for a, b, q, n in rows if b > t0 and a < t1]
T = t1 - t0
print("window %.2f ms, %d steps → %.1f us per step" % (T / 1e6, steps, T / 1e3 / steps))
ev = []
for a, b, q, n in win:
    ev.append((a, 1)); ev.append((b, -1))
ev.sort()
conc = collections.Counter(); cur = 0; last = t0
for t, dlt in ev:
    conc[cur] += t - last; last = t; cur += dlt
conc[cur] += t1 - last
print("concurrency: " + "  ".join("%d: %.1f%%" % (k, 100.0 * v / T) for k, v in sorted(conc.items())), " mean %.2f" % (sum(k * v for k, v in conc.items()) / T))
byq = collections.defaultdict(list)
for a, b, q, n in win:
    byq[q].append((a, b, n))
for q, lst in sorted(byq.items()):
    lst.sort()
    busy = sum(b - a for a, b, n in lst)
    gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
    small = [g for g in gaps if 0 <= g < 20000]
    names = collections.Counter(n for _, _, n in lst).most_common(4)
    print("queue %s: busy %.1f%%, %d kernels, gaps<20us: n=%d mean %.1f us; gaps>=20us total %.1f%% | %s" % (q, 100.0 * busy / T, len(lst), len(small), (sum(small) / max(len(small), 1)) / 1e3, 100.0 * sum(g for g in gaps if g >= 20000) / T, names))
# exclusive time per kernel name
pts = sorted(set([a for a, b, q, n in win] + [b for a, b, q, n in win]))
active = []
excl = collections.Counter(); tot = collections.Counter()
import bisect
starts = sorted(win)
# sweep
evs = sorted([(a, 0, i) for i, (a, b, q, n) in enumerate(win)] + [(b, 1, i) for i, (a, b, q, n) in enumerate(win)])
live = set(); last = t0
for t, kind, i in evs:
    if live:
        if len(live) == 1:
            excl[win[next(iter(live))][3]] += t - last
        for j in live:
            tot[win[j][3]] += (t - last) / len(live)
    last = t
    if kind == 0: live.add(i)
    else: live.discard(i)
print("per kernel (us per step): duration | alone on the GPU | time-share (duration / concurrency)")
dur = collections.Counter()
for a, b, q, n in win: dur[n] += b - a
for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:24]:
    print("   %-20s %8.1f %8.1f %8.1f" % (n, dur[n] / 1e3 / steps, excl[n] / 1e3 / steps, v / 1e3 / steps))
