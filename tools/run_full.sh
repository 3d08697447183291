# Run ON THE GPU BOX (via gpurun): whole GPU suite, then a seq2seq kernel trace.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/full
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/full/pytest.log 2>&1; tail -3 gpurun_out/full/pytest.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/seqtr
FF_SEQ_ONLY_A=1 rocprofv3 --kernel-trace --stats -d /tmp/seqtr -o s -- python tools/time_seq2seq.py > gpurun_out/full/trace.log 2>&1
python tools/rocpd_stats.py /tmp/seqtr/s_results.db gpurun_out/full/kernel_stats.md > /dev/null 2>&1
head -14 gpurun_out/full/kernel_stats.md | cut -c1-150
