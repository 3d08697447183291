#!/bin/bash
# Run ON THE GPU BOX: kernel trace of config B in a given arithmetic form; per-kernel stats + per-step breakdown
# usage: tools/run_r04_trace.sh <tag> <bench args...>
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04
mkdir -p $OUT
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs $*"
rm -rf /tmp/prof_trace
rocprofv3 --kernel-trace --stats -d /tmp/prof_trace -o t -- $CMD > $OUT/trace_$TAG.log 2>&1
python tools/rocpd_stats.py /tmp/prof_trace/t_results.db $OUT/kernel_stats_$TAG.md > /dev/null
python tools/step_breakdown.py /tmp/prof_trace/t_results.db 12,24,36 > $OUT/steps_$TAG.txt 2>&1
tail -1 $OUT/trace_$TAG.log | cut -c1-300
head -30 $OUT/kernel_stats_$TAG.md
cat $OUT/steps_$TAG.txt | head -120
