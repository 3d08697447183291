# Run ON THE GPU BOX (via gpurun): config B with the wireframe's 256 sequences split into groups on concurrent streams.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/split
runb() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs --steps 8 --warmup 3 "$@" 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%.2f' % d['ms_per_step'])"; }
{
echo "one group: $(runb) ms"
echo "2 groups x 128 on 2 streams: $(runb --chunk-seqs 128 --streams 2) ms"
echo "4 groups x 64 on 4 streams: $(runb --chunk-seqs 64 --streams 4) ms"
echo "2 groups x 128 on 1 stream: $(runb --chunk-seqs 128 --streams 1) ms"
} 2>&1 | tee gpurun_out/split/ab.txt
