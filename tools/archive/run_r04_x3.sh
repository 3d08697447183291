# Run ON THE GPU BOX: round-4 3 x bf16 kernel -- op tests, then both launch shapes over the decode's row counts
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "x3 or split_weight" > gpurun_out/r04/x3_tests.log 2>&1; tail -5 gpurun_out/r04/x3_tests.log
timeout 600 python tools/bench_gemm.py --ts 8,12,16,17,20,24,28,32,33,36,64 --tiles 7 --x3-variants > gpurun_out/r04/gemm_x3_variants.txt 2>&1
cat gpurun_out/r04/gemm_x3_variants.txt
