cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ub
./build_ub/mfma_chain > gpurun_out/ub/mfma_chain.txt 2>&1
./build_ub/mfma_lds > gpurun_out/ub/mfma_lds.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d /tmp/vn -o v -- python tools/vendor_names.py > gpurun_out/ub/vn.log 2>&1
python tools/rocpd_stats.py /tmp/vn/v_results.db gpurun_out/ub/vendor_kernels.md > /dev/null 2>&1
rocm-smi --showclocks > gpurun_out/ub/clocks.txt 2>&1
cat gpurun_out/ub/mfma_chain.txt gpurun_out/ub/mfma_lds.txt
