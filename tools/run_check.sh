# Run ON THE GPU BOX (via gpurun): the whole GPU test suite, then the default bench line.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/check
rm -f gpurun_out/check/parity_margins.txt
FF_PARITY_MARGINS=$PWD/gpurun_out/check/parity_margins.txt timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/check/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/check/pytest.log
tail -4 gpurun_out/check/pytest.log
timeout 900 python bench.py "$@" > gpurun_out/check/bench.json 2> gpurun_out/check/bench.err; tail -c 600 gpurun_out/check/bench.json
