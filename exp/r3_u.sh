#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3u
python exp/stamps2.py 64 hdl64 2>&1 | head -11; python exp/stamps2.py 32 agg10 2>&1 | head -11; python exp/stamps2.py 64 hdl64_urban 2>&1 | head -11
python -m pytest tests -m gpu -x -q > gpurun_out/r3u/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r3u/pytest.log
for W in hdl64_b64 os128_b64 agg10_b32 hdl64_urban_b64; do python exp/quick.py $W --workload $W --steps 40 --reps 5 2>gpurun_out/r3u/$W.err | tail -1 | cut -c1-150; done
