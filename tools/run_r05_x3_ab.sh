# Run ON THE GPU BOX: the package-default line of config B (split products) -- the K-piece knob for launches below one tile per CU
# (FF_X3_SMALL_SPLIT) and the row threshold from which the split kernel takes the projections (x3_min_rows), alternating.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
run() { timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --no-x3-line --no-roofline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f' % d['ms_per_step'])"; }
{
for rep in 1 2 3 4; do
  echo "FF_X3_SMALL_SPLIT=0 x3_min_rows=1024: $(FF_X3_SMALL_SPLIT=0 run --x3-min-rows 1024) ms"
  echo "FF_X3_SMALL_SPLIT=1 x3_min_rows=1024: $(FF_X3_SMALL_SPLIT=1 run --x3-min-rows 1024) ms"
done
for rep in 1 2 3; do
  for m in 768 1024 1280 1536 2048; do echo "x3_min_rows=$m: $(run --x3-min-rows $m) ms"; done
done
} | tee gpurun_out/r05/x3_small_end_ab.txt
