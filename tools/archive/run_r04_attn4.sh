# Run ON THE GPU BOX: K/V-resident cross-attention, the blocks of a pair load K / V in rotated order (default) vs the same order
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k attention 2>&1 | tail -3
for k in 1 0 1 0; do
  r=$(FF_RK_ROTATE=$k timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-x3-line 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_time_ms_per_step']['attention_kernels'])")
  echo "FF_RK_ROTATE=$k -> ms_per_step, attention ms: $r"
done | tee gpurun_out/r04/attn_rotate.txt
