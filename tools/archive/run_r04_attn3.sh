# Run ON THE GPU BOX: K/V-resident cross-attention, software-pipelined inside the wave (in-tree) vs the round-3 form
# (build_ub/lib_attn_base.so); attention op tests and the golden parity tests on the in-tree library first
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k attention 2>&1 | tail -3
timeout 900 python -m pytest tests/test_parity_golden.py -m gpu -q -x 2>&1 | tail -3
for lib in "" build_ub/lib_attn_pipe.so build_ub/lib_attn_base.so "" build_ub/lib_attn_pipe.so build_ub/lib_attn_base.so; do
  r=$(FF_HIP_LIB=$lib timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-x3-line 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_time_ms_per_step']['attention_kernels'])")
  echo "lib=${lib:-in-tree} -> ms_per_step, attention ms: $r"
done | tee gpurun_out/r04/attn_qprefetch.txt
