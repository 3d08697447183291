#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel trace (rocprofv3 --kernel-trace --stats) of another bench configuration.
# usage: bash tools/trace_config.sh <name> <bench.py args...>   ->  gpurun_out/profiles_extra/kernel_stats_<name>.md
set -u
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_extra; mkdir -p $OUT
rm -rf /tmp/tr_$NAME
rocprofv3 --kernel-trace --stats -d /tmp/tr_$NAME -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs "$@" > $OUT/trace_$NAME.log 2>&1
python tools/rocpd_stats.py /tmp/tr_$NAME/t_results.db $OUT/kernel_stats_$NAME.md > /dev/null
tail -1 $OUT/trace_$NAME.log | cut -c1-300 > $OUT/bench_line_under_trace_$NAME.txt
rm -rf /tmp/tr_$NAME
head -14 $OUT/kernel_stats_$NAME.md | cut -c1-130
