set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2c
timeout 900 python tools/bench_gemm.py --ts 16,36,64,128 --tiles 2,3,4,5,7 > gpurun_out/r2c/gemm_tiles.txt 2>&1; cat gpurun_out/r2c/gemm_tiles.txt
