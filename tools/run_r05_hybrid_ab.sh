# Run ON THE GPU BOX: the hybrid launch of the 64x64 f32 family (FF_SK_HYBRID=1, the default) against unit ranges / whole tiles
# (FF_SK_HYBRID=0): op tests, the projection table at the staircase steps, config B alternating.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
if [ -z "${FF_AB_SKIP_TESTS:-}" ]; then
timeout 1200 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_parity_golden.py -m gpu -q -x -k "f32_matrix_cores_only or (golden_parity and not bf16)" 2>&1 | tail -2
fi
{
for h in 0 1; do
  echo "## FF_SK_HYBRID=$h  (tile 7 = the automatic choice)"
  FF_SK_HYBRID=$h timeout 600 python tools/bench_gemm.py --ts ${FF_AB_TS:-8,9,10,12,14,17,20,25,33} --tiles 7 2>&1 | grep -v "^/opt"
done
for fix in ${FF_AB_FIXES:-10}; do
for rep in 1 2 3 4 5 6; do
  for h in 0 1; do
    r=$(FF_SK_HYBRID=$h FF_SK_HYBRID_FIX=$fix timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --no-x3-line 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f %.3f' % (d['ms_per_step'], d['kernel_time_ms_per_step']['gemm_f32_kernels']))")
    echo "FF_SK_HYBRID=$h fix=$fix -> ms_per_step, gemm ms: $r"
  done
done
done
} | tee gpurun_out/r05/gemm_hybrid_ab.txt
