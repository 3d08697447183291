# Run ON THE GPU BOX: whole GPU suite (margins file), then a kernel trace of config A (one single-sequence wireframe) with the
# kernel-by-kernel listing of steps 10 / 100 / 250.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06f; mkdir -p $O
rm -f $O/parity_margins.txt
FF_PARITY_MARGINS=$PWD/$O/parity_margins.txt timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_full.log 2>&1; tail -5 $O/pytest_gpu_full.log
rm -rf /tmp/prof_trace
FF_SEQ_ONLY_A=1 FF_SEQ_REPS=2 rocprofv3 --kernel-trace --stats -d /tmp/prof_trace -o t -- python tools/time_seq2seq.py > $O/trace_A.log 2>&1
python tools/rocpd_stats.py /tmp/prof_trace/t_results.db $O/kernel_stats_A1.md > /dev/null
python tools/step_breakdown.py /tmp/prof_trace/t_results.db 10,100,250 > $O/steps_A1.txt 2>&1
tail -3 $O/trace_A.log; head -16 $O/kernel_stats_A1.md; grep -A52 "step 100" $O/steps_A1.txt | head -60
