# Run ON THE GPU BOX: 2 x fp16 split products -- LayerNorm applied first (MODE 1) or in the epilogue (MODE 3): config B, C128, E32.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06e; mkdir -p $O
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
x=d.get("package_default", d.get("bf16x3_projections", {}))
print("  B split %.2f ms %.0f edges/s" % (x.get("ms_per_step",-1), x.get("value",-1)), {k:(v.get("default") or v.get("bf16x3") or {}).get("value") for k,v in (d.get("other_configs") or {}).items()})
PY
}
{
for i in 1 2; do
  echo "LayerNorm in the epilogue (MODE 3):"; FF_BENCH_LN_EPILOGUE=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 6 --warmup 2 --other-list C128,E32 2>/dev/null > $O/b_epi.json; show $O/b_epi.json
  echo "LayerNorm first (MODE 1):"; FF_BENCH_LN_EPILOGUE=0 timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 6 --warmup 2 --other-list C128,E32 2>/dev/null > $O/b_first.json; show $O/b_first.json
done
} > $O/fp16x2_ln_form_ab.txt 2>&1
cat $O/fp16x2_ln_form_ab.txt
