#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3t
python exp/e2e_probe.py 2>&1 | tail -6
python -m pytest tests -m gpu -x -q -k "blob or async or edge or errors or two_batches" 2>&1 | tail -3
