# Run ON THE GPU BOX: K/V-resident cross-attention by the older/younger waves' item shares (config B f32 line, attention op tests)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
for sp in "1 1" "10 8" "11 7" "12 6" "5 4"; do
  set -- $sp
  r=$(FF_RK_SPLIT_OLD=$1 FF_RK_SPLIT_YOUNG=$2 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-x3-line 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_time_ms_per_step']['attention_kernels'])")
  echo "old:young $1:$2 -> ms_per_step, attention ms: $r"
done | tee gpurun_out/r04/attn_split.txt
FF_RK_SPLIT_OLD=11 FF_RK_SPLIT_YOUNG=7 timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k attention 2>&1 | tail -3
