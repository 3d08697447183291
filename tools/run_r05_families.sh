# Run ON THE GPU BOX: the two f32 projection families side by side on the decode's shapes, 2 k ... 9 k rows, in the LayerNorm-folded
# forms the decode uses (tools/bench_gemm_x3_ln.py: column "f32 fold" = the automatic choice, "dma fold" = the LDS-DMA kernel forced)
# and in the plain form (tools/bench_gemm.py, tile 7 = automatic).  FF_DMA_MIN_ROWS=100000000 keeps the automatic choice inside the
# 64x64 family (persistent / stream-K / hybrid), the per-width thresholds are switched off the same way.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
{
echo "## LayerNorm-folded forms; automatic choice = the 64x64 family (FF_DMA_MIN_ROWS=1e8): compare columns 'f32 fold' (64x64 family) and 'dma fold' (LDS-DMA kernel)"
FF_DMA_MIN_ROWS=100000000 FF_DMA_MIN_ROWS_N512=100000000 FF_DMA_MIN_ROWS_WIDE=100000000 timeout 600 python tools/bench_gemm_x3_ln.py --ms 2048,2304,2560,2816,3072,3328,3584,3840,4096,4352,5120,6144,6400,7680,8448,9216 2>&1 | grep -v "^/opt" | cut -c1-36,64-100
echo "## plain form, tile 7 automatic inside the 64x64 family (FF_DMA_MIN_ROWS=1e8), TF/s"
FF_DMA_MIN_ROWS=100000000 FF_DMA_MIN_ROWS_N512=100000000 FF_DMA_MIN_ROWS_WIDE=100000000 timeout 600 python tools/bench_gemm.py --ts 16,17,18,20,22,24,25,26 --tiles 7 2>&1 | grep -v "^/opt" | cut -c1-34
echo "## plain form, LDS-DMA kernel (tile 11) on the same shapes, TF/s"
timeout 600 python tools/bench_gemm.py --ts 16,17,18,20,22,24,25,26 --tiles 11 2>&1 | grep -v "^/opt" | cut -c1-34
} > gpurun_out/r05/gemm_families_4k_9k.txt 2>&1
tail -5 gpurun_out/r05/gemm_families_4k_9k.txt
