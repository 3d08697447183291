# Run ON THE GPU BOX: the f32 LDS-DMA kernel (tile 11) -- op tests, then its rate beside the automatic choice (tile 7) and the split kernel
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "dma_kernel or layernorm_segment or layernorm_statistics" > gpurun_out/r04/dma_tests.log 2>&1; tail -5 gpurun_out/r04/dma_tests.log
timeout 600 python tools/bench_gemm.py --ts 4,8,12,16,17,20,24,28,32,33,36,64,128 --tiles 7,11 > gpurun_out/r04/gemm_dma_f32.txt 2>&1
cat gpurun_out/r04/gemm_dma_f32.txt
