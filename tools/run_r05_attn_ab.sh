# Run ON THE GPU BOX: an attention variant build (FF_HIP_LIB=build_ub/lib_<name>.so) against the in-tree library:
# attention op tests + the golden parity tests on the variant first, then config B alternating (ms per step, attention ms).
# usage: bash tools/run_r05_attn_ab.sh <name> [<name2> ...]
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
for n in "$@"; do
  FF_HIP_LIB=$PWD/build_ub/lib_$n.so timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k attention 2>&1 | tail -2
  FF_HIP_LIB=$PWD/build_ub/lib_$n.so timeout 1200 python -m pytest tests/test_parity_golden.py -m gpu -q -x -k "golden_parity and not bf16" 2>&1 | tail -2
done
for rep in 1 2 3; do
  for n in "" "$@"; do
    lib=""; [ -n "$n" ] && lib=$PWD/build_ub/lib_$n.so
    r=$(FF_HIP_LIB=$lib timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --no-x3-line 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f %.3f' % (d['ms_per_step'], d['kernel_time_ms_per_step']['attention_kernels']))")
    echo "lib=${n:-in-tree} -> ms_per_step, attention ms: $r"
  done
done | tee gpurun_out/r05/attn_${1}_ab.txt
