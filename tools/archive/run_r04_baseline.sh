# Run ON THE GPU BOX: round-4 baselines (bf16 MFMA pattern rates, projection kernels over the decode's row counts)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
./build_ub/mfma_bf16 > gpurun_out/r04/ubench_mfma_bf16.txt 2>&1
python tools/bench_gemm.py --ts 4,8,12,16,20,24,25,28,32,33,36,64 --tiles 7 > gpurun_out/r04/gemm_base.txt 2>&1
tail -50 gpurun_out/r04/gemm_base.txt; cat gpurun_out/r04/ubench_mfma_bf16.txt
