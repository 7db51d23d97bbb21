#!/bin/bash
set -u
O=gpurun_out/r2c10; mkdir -p $O
export TMPDIR=/tmp
echo "== pytest all gpu"; timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo rc=$?; tail -40 $O/pytest.log
