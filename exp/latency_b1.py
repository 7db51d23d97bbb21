import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicslamtool_amd import engine, kitti_params, synth
p = kitti_params(1)
b = engine.MorBatch(p, 1, 120000)
frames = [synth.frame(2005, "hdl64", f) for f in range(12)]
hin = engine.HostBuffer((120000, 4)); hout = engine.HostBuffer((120000, 4))
for mode in ("pageable", "pinned", "device"):
    buf = engine.DeviceBuffer(120000 * 16)
    ts = []
    for f, (x, pose) in enumerate(frames):
        t0 = time.perf_counter()
        if mode == "pageable":
            b.push([x], pose[None, :]); out = b.filter()
        elif mode == "pinned":
            hin.array[...] = x
            t0 = time.perf_counter()
            b.push([hin.array], pose[None, :]); b.filter_into([hout.array])
        else:
            buf.upload(x)
            t0 = time.perf_counter()
            b.push([(buf, 120000)], pose[None, :]); b.filter(to_host=False)
        ts.append(time.perf_counter() - t0)
    print("%-9s push+filter latency of one 120 000-point cloud: median %.3f ms (min %.3f)" % (mode, 1e3 * np.median(ts[2:]), 1e3 * min(ts[2:])))
    buf.free()
