cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "" nosplit mixsplit; do
  echo "== variant: ${v:-in-tree}"
  if [ -n "$v" ]; then export FF_HIP_LIB=$PWD/build_ub/lib_$v.so; else unset FF_HIP_LIB; fi
  python tools/bench_split_kinds.py 9,36,128 2>&1 | grep -v amdgpu.ids
done
export FF_HIP_LIB=$PWD/build_ub/lib_mixsplit.so
timeout 300 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "x2h" 2>&1 | tail -3
