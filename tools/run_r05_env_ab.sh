# Run ON THE GPU BOX: config B (f32 line) alternating between two settings of ONE environment variable.
# usage: bash tools/run_r05_env_ab.sh VAR valueA valueB [reps] [extra bench args...]
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
var=$1; a=$2; b=$3; reps=${4:-5}; shift 4 || shift $#
for rep in $(seq 1 $reps); do
  for v in $a $b; do
    r=$(env $var=$v timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --no-x3-line "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f %.3f %.3f' % (d['ms_per_step'], d['kernel_time_ms_per_step']['gemm_f32_kernels'], d['kernel_time_ms_per_step']['attention_kernels']))")
    echo "$var=$v -> ms_per_step, gemm ms, attention ms: $r"
  done
done | tee gpurun_out/r05/env_ab_${var}.txt
