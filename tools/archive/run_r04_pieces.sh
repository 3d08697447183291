# Run ON THE GPU BOX: projection kernels -- round-4 code (build_ub/lib_x3_head.so) vs the shared exchange loops (in-tree, fewer
# registers) vs the same with the f32 LDS-DMA kernel at four blocks per CU (build_ub/lib_x3_c.so); FF_X3_PIECES2=0 throughout
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
export FF_X3_PIECES2=0
timeout 1500 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm_x3 or gemm_dma or ln" 2>&1 | tail -3
FF_HIP_LIB=build_ub/lib_x3_c.so timeout 1500 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm_dma or ln" 2>&1 | tail -3
for lib in build_ub/lib_x3_head.so "" build_ub/lib_x3_c.so; do
  echo "== lib=${lib:-in-tree}"
  FF_HIP_LIB=$lib timeout 600 python tools/bench_gemm.py --batch 256 --ts 16,24,36,128 --tiles 11
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/gemm_regs_ab.txt
for lib in build_ub/lib_x3_head.so "" build_ub/lib_x3_c.so build_ub/lib_x3_head.so "" build_ub/lib_x3_c.so; do
  r=$(FF_HIP_LIB=$lib timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['bf16x3_projections']['ms_per_step'])")
  echo "lib=${lib:-in-tree} -> config B ms per wireframe: f32, package default: $r"
done | tee gpurun_out/r04/regs_ab.txt
