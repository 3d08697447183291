set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/trace
rm -rf /tmp/prof_trace
rocprofv3 --kernel-trace --stats -d /tmp/prof_trace -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs $* > gpurun_out/trace/trace.log 2>&1
python tools/rocpd_stats.py /tmp/prof_trace/t_results.db gpurun_out/trace/kernel_stats.md > /dev/null
python tools/step_breakdown.py /tmp/prof_trace/t_results.db 4,8,20,36 > gpurun_out/trace/steps.txt 2>&1
head -45 gpurun_out/trace/steps.txt
