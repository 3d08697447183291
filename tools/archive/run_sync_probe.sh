# Run ON THE GPU BOX (via gpurun): config B by the period of the host's stop-rule check (sync_every), never-stopping weights:
# what the check itself costs.
cd "$GRAFT_REPO_ROOT"
runb() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs --steps 8 --warmup 3 "$@" 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%.2f' % d['ms_per_step'])"; }
for i in 1 2; do for se in 4 2 1 0; do echo "sync_every=$se: B $(runb --sync-every $se) ms"; done; done
