"""One kernel's duration alone (synchronous steps) under the library named by MOR_HIP_LIB — results of a cut build are wrong, only the duration is read (exp/cutk.sh's method, lighter).
usage: MOR_HIP_LIB=… python exp/cut_time.py <workload> <kernel> [steps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dynamicslamtool_amd import engine, kitti_params, synth, shard
wl, kern = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
_, _, _, mo, go = bench.WORKLOADS[wl]
p = kitti_params(mo or 1)
p.ground_method = go if go is not None else 0
engine.bind_thread_to_device_node(0)
leg = bench.Leg(engine, synth, shard, p, wl, 0, 0, 6)
try:
    for _ in range(3):
        leg.step()
    al = leg.kernel_leg(steps, sync=True)
    print(json.dumps({"lib": os.path.basename(os.environ.get("MOR_HIP_LIB", "in-tree")), "workload": wl, kern: al.get(kern, {}).get("avg_us"), "k_cg_final": al.get("k_cg_final", {}).get("avg_us")}))
except Exception as e:
    print(json.dumps({"lib": os.path.basename(os.environ.get("MOR_HIP_LIB", "in-tree")), "error": repr(e)[:200]}))
