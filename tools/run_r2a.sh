set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -25 gpurun_out/r2a/pytest.log
