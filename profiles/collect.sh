#!/bin/bash
# Collects the round's profiles on the GPU box (run through gpurun from the repo root: bash profiles/collect.sh r03):
#   1. rocprofv3 --kernel-trace --stats of the bench command, headline workload           → profiles/rNN_kernel_stats.csv
#      (+ the same for the other workloads of bench.py's WORKLOADS table                   → profiles/rNN_kernel_stats_<name>.csv)
#   2. per workload two separate PMC passes (FETCH_SIZE, WRITE_SIZE — they do not fit one pass on gfx950)
#      → per-kernel HBM traffic per launch → profiles/traffic_<workload>.json (read back by bench.py → roofline.per_kernel / path traffic)
#   3. SQ and TCC counters of the headline workload                                        → profiles/rNN_counters.json
# HBM bytes = 2 × FETCH_SIZE·1024 + WRITE_SIZE·1024: on gfx950 FETCH_SIZE counts half the bytes of wide coalesced
# reads (MI355X_MICROARCH.md §HBM); other access widths are uncalibrated, so the figure is an estimate for the
# scattered 16-byte accesses of the grid build / scoring kernels.
# Counter passes never combine --pmc with sys/hip/hsa tracing (only --kernel-trace), and the profiled program follows `--` directly.
set -e
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m dynamicslamtool_amd.build   # BEFORE the first rocprofv3 line: no compiler may start under the profiler's preload (engine.lib() refuses to autobuild there and fails loudly on a stale library)
export MOR_NO_AUTOBUILD=1
OUT=gpurun_out/prof_$R
rm -rf $OUT; mkdir -p $OUT profiles
HEAD_ID=$(cat .git_head 2>/dev/null || echo unknown)
BASE="--no-cpu-baseline --no-kernel-timing --no-extras"
WL="os128_b64 agg10_b32 hdl64_urban_b64 hdl64_b64_method2 hdl64_b64_voxel_ground"
python3 bench.py --no-cpu-baseline --no-extras --detail $OUT/bench_untraced_detail.json > $OUT/bench_untraced.json 2> $OUT/bench_untraced.err || true   # the same leg without the tracer, bench.py's own HIP-event timing on
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py $BASE --detail $OUT/bench_trace_detail.json > $OUT/bench_trace.json 2> $OUT/trace.err
cp $OUT/trace/t_kernel_stats.csv profiles/${R}_kernel_stats.csv
for W in $WL; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -o t -- python3 bench.py --workload $W --steps 20 --warmup 3 $BASE --detail $OUT/bench_${W}_detail.json > $OUT/bench_$W.json 2> $OUT/trace_$W.err || true
  cp $OUT/trace_$W/t_kernel_stats.csv profiles/${R}_kernel_stats_$W.csv 2>/dev/null || true
done
# counter passes serialise the kernels and cost about a second per dispatch: they profile exp/pmc_run.py (three synchronous steps of the workload, nothing else);
# the first step (no previous frame: no pair stage) is part of the per-launch averages of the frame-independent kernels only
for W in hdl64_b64 $WL; do
  PMC="python3 exp/pmc_run.py $W 3"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$W -o f -- $PMC > /dev/null 2> $OUT/fetch_$W.err || true
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$W -o w -- $PMC > /dev/null 2> $OUT/write_$W.err || true
done
PMC="python3 exp/pmc_run.py hdl64_b64 3"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/sq -o s -- $PMC > /dev/null 2> $OUT/sq.err || true
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq2 -o s -- $PMC > /dev/null 2> $OUT/sq2.err || true
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/tc -o t -- $PMC > /dev/null 2> $OUT/tc.err || true
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT/sq3 -o s -- $PMC > /dev/null 2> $OUT/sq3.err || true
python3 profiles/summarise.py "$R" "$OUT" "$HEAD_ID"
cp profiles/traffic_*.json profiles/${R}_summary.md profiles/${R}_kernel_stats*.csv profiles/${R}_counters.json gpurun_out/ 2>/dev/null || true
