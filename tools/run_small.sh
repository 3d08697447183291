set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2c
timeout 900 python tools/bench_gemm.py --ts 1,2,3,4,6 --tiles 3,6,8 > gpurun_out/r2c/gemm_small.txt 2>&1; cat gpurun_out/r2c/gemm_small.txt
