set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2e
run() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-x3-line --steps 8 --warmup 2 "$@" 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%.2f' % d['ms_per_step'])"; }
echo "base $(run) $(run)"
for g in 2,1024,25,1024 2,4096,25,1024 2,8192,25,1024 2,2048,10,1024 2,2048,40,1024 4,2048,25,1024 2,2048,25,768 2,2048,25,1536 2,2048,25,512; do
  echo "tuning $g: $(run --gemm-tuning $g)"
done
echo "base $(run)"
echo "sync0 $(run --sync-every 0)"
echo "attn1 $(run --attn-algo 1)  attn2 $(run --attn-algo 2)"
