# Run ON THE GPU BOX: the 2 x fp16 cross-attention kernel -- op tests (every attention test with algo 4), goldens of the split forms,
# config B / C128 / E32 package-default lines with and without it (FF_X2H_ATTN).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06h; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "attention" > $O/pytest_attention.log 2>&1; tail -6 $O/pytest_attention.log
if [ "${1:-all}" = "ops" ]; then exit 0; fi
rm -f $O/parity_margins_split.txt
FF_PARITY_MARGINS=$PWD/$O/parity_margins_split.txt timeout 1500 python -m pytest tests/test_parity_golden.py -m gpu -q -x -k "split_projections and fp16x2" > $O/pytest_split_goldens.log 2>&1; tail -4 $O/pytest_split_goldens.log
show() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
x=d.get("package_default",{})
print("  B default %.2f ms %.0f edges/s" % (x.get("ms_per_step",-1), x.get("value",-1)), {k:(v.get("default") or {}).get("value") for k,v in (d.get("other_configs") or {}).items()})
PY
}
{
for i in 1 2; do
  echo "f32 attention kernels (FF_X2H_ATTN=0):"; FF_X2H_ATTN=0 timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-live-traffic --steps 6 --warmup 2 --other-list C128,E32 2>/dev/null > $O/b_off.json; show $O/b_off.json
  echo "2 x fp16 cross-attention (default):"; timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-live-traffic --steps 6 --warmup 2 --other-list C128,E32 2>/dev/null > $O/b_on.json; show $O/b_on.json
done
} > $O/x2h_attention_ab.txt 2>&1
cat $O/x2h_attention_ab.txt
