"""Latency through the drop-in class (mor_replay over .bin frames of one stream) under the adapter's input / output modes, next to the C-ABI latency from page-locked memory.
usage (through gpurun): python exp/class_latency.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
res = {}
for name, env in (("default", {}), ("in=pageable,out=staged (rounds 1-5)", {"MOR_CLASS_INPUT": "pageable", "MOR_CLASS_OUTPUT": "staged"}), ("in=pageable", {"MOR_CLASS_INPUT": "pageable"}),
                  ("in=zerocopy", {"MOR_CLASS_INPUT": "zerocopy"}), ("out=staged", {"MOR_CLASS_OUTPUT": "staged"})):
    res[name] = bench.class_latency(env_extra=env)
res["visualize build, default"] = bench.class_latency(binary="mor_replay")
res["latency_b1_ms (C ABI, page-locked)"] = {w: bench.latency_b1(w, 0) for w in ("hdl64_b64", "hdl64_urban_b64")}
print(json.dumps(res, indent=1))
