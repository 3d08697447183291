# Run ON THE GPU BOX: config B / C128 f32 lines by the row threshold of the LDS-DMA kernel and the LayerNorm folding limit
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
for thr in 1099511627776 9000 7168 5120 4096; do
  r=$(FF_DMA_MIN_ROWS=$thr timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-x3-line --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "config B f32, DMA kernel from $thr rows: $r ms"
done | tee gpurun_out/r04/dma_threshold_ab.txt
for thr in 1099511627776 7168; do for fm in 0 1073741824; do
  r=$(FF_DMA_MIN_ROWS=$thr timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --other-list C128 --other-steps 2 --no-x3-line --no-roofline --ln-fuse-max-rows $fm 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['other_configs']['C128']['ms_per_step'], d['other_configs']['C128']['value'])")
  echo "C128 f32, DMA kernel from $thr rows, ln_fuse_max_rows $fm: $r"
done; done | tee -a gpurun_out/r04/dma_threshold_ab.txt
