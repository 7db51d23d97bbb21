#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3g
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kernel_variants or voxel_covariance or edge_cases or blob" > gpurun_out/r3g/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r3g/pytest.log
run() { tag=$1; shift; env "$@" python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r3g/$tag.json 2> gpurun_out/r3g/$tag.err; python - $tag <<'PY'
import json,sys
t=sys.argv[1]
try:
    d=json.loads(open("gpurun_out/r3g/%s.json"%t).read().strip().splitlines()[-1]); print(t, d["value"], d["ms_per_step"], d["sanity"]["ok"], "sum", d["roofline"]["sum_kernel_us_per_step_pipelined"], d["roofline"]["sum_kernel_us_per_step_alone"])
    for k in ("k_split","k_classify","k_scatter"):
        if k in d["kernels"]: print("    ", k, d["kernels"][k]["avg_us"], d["kernels_alone_avg_us"].get(k))
except Exception as e: print(t,"fail",e); print(open("gpurun_out/r3g/%s.err"%t).read()[-800:])
PY
}
run g8 MOR_SINGLE_PASS_SPLIT=1
run g4 MOR_SINGLE_PASS_SPLIT=1 MOR_HIP_LIB=$PWD/exp/libmor_spg4.so
run g16 MOR_SINGLE_PASS_SPLIT=1 MOR_HIP_LIB=$PWD/exp/libmor_spg16.so
run two MOR_SINGLE_PASS_SPLIT=0
run g8b MOR_SINGLE_PASS_SPLIT=1
