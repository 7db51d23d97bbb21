import os, sys, time, numpy as np, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicslamtool_amd import engine
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in glob.glob("/sys/class/drm/card*/device/numa_node"): print(p, open(p).read().strip())
for p in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")): print(p, open(p).read().strip())
print("device_count", engine.device_count())
n = 123 << 20
dev = engine.DeviceBuffer(n)
def rate(tag):
    h = engine.HostBuffer((n,), np.uint8)
    L = engine.lib()
    for _ in range(2): L.mor_device_download(0, h.ptr, dev.ptr, n)
    t = time.perf_counter(); 
    for _ in range(5): L.mor_device_download(0, h.ptr, dev.ptr, n)
    d2h = 5 * n / (time.perf_counter() - t) / 1e9
    t = time.perf_counter(); 
    for _ in range(5): L.mor_device_upload(0, dev.ptr, h.ptr, n)
    h2d = 5 * n / (time.perf_counter() - t) / 1e9
    print("%-28s cpu %3d  D2H %.1f GB/s  H2D %.1f GB/s" % (tag, os.sched_getcpu() if hasattr(os, "sched_getcpu") else -1, d2h, h2d))
    h.free()
rate("fresh")
cores = sorted(os.sched_getaffinity(0))
for c in (cores[0], cores[len(cores) // 4], cores[len(cores) // 2], cores[3 * len(cores) // 4], cores[-1]):
    os.sched_setaffinity(0, {c}); rate("allocated on core %d" % c)
