#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the round-6 profile set -> gpurun_out/profiles_r06/ (copied to profiles/r06/ afterwards).
#   kernel trace + the five PMC passes of the default bench command (f32 headline) and of the package-default form (--x3-min-rows
#   1024 = two-term fp16 split products), per-step breakdowns, MFMA-busy, the effective-clock table, kernel traces of C128 / E32 in
#   both forms and of the seq2seq batch (A64), seq2seq timings, the default bench line.
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_r06
mkdir -p $OUT
if [ "${1:-all}" = "clock" ] || [ "${1:-all}" = "all" ]; then
  timeout 600 python tools/effective_clock.py > $OUT/effective_clock.md 2> $OUT/effective_clock.err
fi
if [ "${1:-all}" = "pmc" ] || [ "${1:-all}" = "all" ]; then
  bash tools/collect_profiles.sh r06 > $OUT/collect.log 2>&1
  ( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
    CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs --no-live-traffic --x3-min-rows 1024"
    echo "command: $CMD" > $OUT/command_x3.txt
    rm -rf /tmp/prof_x3 && rocprofv3 --kernel-trace --stats -d /tmp/prof_x3 -o t -- $CMD > $OUT/trace_x3.log 2>&1
    python tools/rocpd_stats.py /tmp/prof_x3/t_results.db $OUT/kernel_stats_x3.md > /dev/null
    python tools/step_breakdown.py /tmp/prof_x3/t_results.db 8,20,33,36 > $OUT/steps_x3.txt 2>&1
    for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
      set -- $pass; name=$1; shift
      rm -rf /tmp/prof_pmc && rocprofv3 --pmc "$@" -d /tmp/prof_pmc -o p -- $CMD > /dev/null 2>&1
      python tools/pmc_per_kernel.py /tmp/prof_pmc/p_results.db $OUT/pmc_${name}_x3.md > /dev/null
    done
    rm -rf /tmp/prof_pmc /tmp/prof_x3 )
  ( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && rm -rf /tmp/prof_trace2 &&
    rocprofv3 --kernel-trace --stats -d /tmp/prof_trace2 -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs --no-live-traffic > /dev/null 2>&1 &&
    python tools/step_breakdown.py /tmp/prof_trace2/t_results.db 4,8,9,20,24,25,32,33,36 > $OUT/steps.txt 2>&1; rm -rf /tmp/prof_trace2 )
  python tools/make_traffic_json.py $OUT > /dev/null 2>&1
  python tools/mfma_busy.py $OUT > $OUT/mfma_busy.txt 2>&1
fi
if [ "${1:-all}" = "configs" ] || [ "${1:-all}" = "all" ]; then
  bash tools/trace_config.sh C128 --wireframes-per-gpu 128 --no-live-traffic > /dev/null 2>&1
  bash tools/trace_config.sh C128_x3 --wireframes-per-gpu 128 --x3-min-rows 1024 --no-live-traffic > /dev/null 2>&1
  bash tools/trace_config.sh E32 --config E --wireframes-per-gpu 32 --no-live-traffic > /dev/null 2>&1
  cp gpurun_out/profiles_extra/kernel_stats_*.md gpurun_out/profiles_extra/bench_line_under_trace_*.txt $OUT/ 2>/dev/null
fi
if [ "${1:-all}" = "seq" ] || [ "${1:-all}" = "all" ]; then
  ( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && rm -rf /tmp/prof_seq &&
    FF_SEQ_ONLY_A=1 FF_SEQ_BATCHES=64 FF_SEQ_REPS=2 rocprofv3 --kernel-trace --stats -d /tmp/prof_seq -o t -- python tools/time_seq2seq.py > $OUT/seq2seq_A64_under_trace.txt 2>&1 &&
    python tools/rocpd_stats.py /tmp/prof_seq/t_results.db $OUT/kernel_stats_A64.md > /dev/null; rm -rf /tmp/prof_seq )
  timeout 600 python tools/time_seq2seq.py > $OUT/seq2seq.txt 2>&1
fi
if [ "${1:-all}" = "bench" ] || [ "${1:-all}" = "all" ]; then
  T0=$(date +%s); timeout 1200 python bench.py > $OUT/bench_line_final.json 2> $OUT/bench_stderr_final.txt; echo "bench.py wall: $(( $(date +%s) - T0 )) s" > $OUT/bench_wall.txt
  cp bench_detail.json $OUT/bench_detail_final.json 2>/dev/null
fi
ls -la $OUT
