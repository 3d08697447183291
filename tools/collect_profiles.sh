#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel trace + PMC passes of the default bench command.
# The rocpd sqlite files are summarised on the box (tools/rocpd_stats.py, tools/pmc_per_kernel.py) and
# deleted; only the small markdown summaries come back under gpurun_out/ (copy them into profiles/).
set -u
TAG=${1:-r01}
shift || true
EXTRA="$*"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs --no-live-traffic $EXTRA"
echo "command: $CMD" > $OUT/command.txt
rocprofv3 --kernel-trace --stats -d /tmp/prof_trace -o t -- $CMD > $OUT/trace.log 2>&1
python tools/rocpd_stats.py /tmp/prof_trace/t_results.db $OUT/kernel_stats.md > /dev/null
tail -1 $OUT/trace.log | cut -c1-400 > $OUT/bench_line_under_trace.txt
pass() {  # name, counters...
  local name=$1; shift
  rm -rf /tmp/prof_pmc
  rocprofv3 --pmc "$@" -d /tmp/prof_pmc -o p -- $CMD > $OUT/pmc_$name.log 2>&1
  python tools/pmc_per_kernel.py /tmp/prof_pmc/p_results.db $OUT/pmc_$name.md > /dev/null
  rm -rf /tmp/prof_pmc
}
# separate PMC passes (TCC: FETCH_SIZE costs 3 slots, WRITE_SIZE 2 -> two passes); no trace domains mixed in
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32
pass l2 TCC_HIT_sum TCC_MISS_sum
rm -f $OUT/pmc_*.log
ls -la $OUT
