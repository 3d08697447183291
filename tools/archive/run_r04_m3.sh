# Run ON THE GPU BOX: split kernel, LayerNorm-in-the-epilogue form -- row statistics requested together with the first operand
# group (in-tree) vs one round trip earlier on their own (build_ub/lib_x3_head.so)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm_x3" 2>&1 | tail -3
for lib in build_ub/lib_x3_head.so "" build_ub/lib_x3_head.so ""; do
  echo "== lib=${lib:-in-tree}"
  FF_HIP_LIB=$lib timeout 600 python tools/bench_gemm_x3_ln.py --ms 4096,9216,32768
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/gemm_m3_epilogue.txt
for lib in build_ub/lib_x3_head.so "" build_ub/lib_x3_head.so ""; do
  r=$(FF_HIP_LIB=$lib timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['bf16x3_projections']['ms_per_step'])")
  echo "lib=${lib:-in-tree} -> config B ms per wireframe: f32, package default: $r"
done | tee gpurun_out/r04/m3_epilogue_ab.txt
