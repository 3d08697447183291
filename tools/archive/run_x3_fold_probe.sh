cd "$GRAFT_REPO_ROOT"
for f in 0 4096 6144; do
python bench.py --no-cpu-baseline --no-roofline --no-other-configs --steps 8 --warmup 3 --ln-fuse-max-rows $f 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('ln_fuse_max_rows $f: f32 %.2f ms, x3 line %.2f ms' % (d['ms_per_step'], d['bf16x3_projections']['ms_per_step']))"
done
