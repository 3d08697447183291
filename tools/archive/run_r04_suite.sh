# Run ON THE GPU BOX: the whole GPU suite on the default library (+ parity margins), no -x
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/parity_margins.txt
FF_PARITY_MARGINS=$PWD/gpurun_out/r04/parity_margins.txt timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r04/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/pytest_gpu.log
grep -n "FAILED\|passed\|failed" gpurun_out/r04/pytest_gpu.log | tail -12
