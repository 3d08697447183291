# Run ON THE GPU BOX: LayerNorm-folding threshold sweep on config B (same box, interleaved).
set -u
cd "$GRAFT_REPO_ROOT"
run() { timeout 600 python bench.py --no-cpu-baseline --no-x3-line --no-other-configs --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%.2f ms/step  gemm %.2f ms (%d launches)  attn %.2f  ln %.2f (%d launches)' % (d['ms_per_step'], d['kernel_time_ms_per_step']['gemm_f32_kernels'], d['kernel_launches_per_step']['gemm_f32_kernels'], d['kernel_time_ms_per_step']['attention_kernels'], d['kernel_time_ms_per_step']['layernorm_kernel'], d['kernel_launches_per_step']['layernorm_kernel']))"; }
for rep in 1 2; do
for r in 4096 6144 8192 1000000; do
echo "ln_fuse_max_rows $r : $(run --ln-fuse-max-rows $r)"
done
done
