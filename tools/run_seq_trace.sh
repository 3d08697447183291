# Run ON THE GPU BOX (via gpurun): seq2seq timings (1 / 8 / 64 wireframes per call) and a kernel trace of config A.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/seq
timeout 900 python tools/time_seq2seq.py > gpurun_out/seq/seq2seq.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/seqtr
FF_SEQ_ONLY_A=1 rocprofv3 --kernel-trace --stats -d /tmp/seqtr -o s -- python tools/time_seq2seq.py > gpurun_out/seq/trace.log 2>&1
python tools/rocpd_stats.py /tmp/seqtr/s_results.db gpurun_out/seq/kernel_stats.md > /dev/null 2>&1
cat gpurun_out/seq/seq2seq.txt; head -24 gpurun_out/seq/kernel_stats.md | cut -c1-150
