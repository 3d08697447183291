# Run ON THE GPU BOX (via gpurun): the LayerNorm-fused projection forms (tools/bench_gemm_ln.py) with variant builds of ff_gemm.hip
# (tools/build_variant.sh: build_ub/lib_<name>.so).  usage: run_ln_variant_probe.sh <name>...   ("" = the in-tree library)
cd "$GRAFT_REPO_ROOT"
for v in "" "$@"; do
  echo "== variant: ${v:-in-tree}"
  if [ -n "$v" ]; then export FF_HIP_LIB=$PWD/build_ub/lib_$v.so; else unset FF_HIP_LIB; fi
  python tools/bench_gemm_ln.py 24,36 2>/dev/null | grep -E "^ +(6144|9216)"
done
