# Run ON THE GPU BOX: K/V-resident cross-attention with the two waves of a SIMD de-phased (priority / start offset)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
for k in 0 1 2 3 4 0; do
  r=$(FF_RK_PHASE=$k timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-x3-line 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_time_ms_per_step']['attention_kernels'])")
  echo "FF_RK_PHASE=$k -> ms_per_step, attention ms: $r"
done | tee gpurun_out/r04/attn_phase.txt
