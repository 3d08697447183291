# Run ON THE GPU BOX (via gpurun): step graphs (FF_GRAPH, opt-in) against plain launches, by steps per graph: config B and config A.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/graph
runb() { timeout 300 python bench.py ${BARGS:-} --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%.2f' % d['ms_per_step'])"; }
runs() { FF_SEQ_ONLY_A=1 timeout 600 python tools/time_seq2seq.py 2>&1 | grep seq2seq | awk '{print $5}' | tr '\n' ' '; }
{
for i in 1 2; do
  echo "no graphs: B $(runb) ms  seq2seq $(runs) ms"
  for k in 1 4 16 1000; do
    echo "graphs of $k steps: B $(BARGS="--graphs 1" FF_GRAPH_STEPS=$k runb) ms  seq2seq $(FF_TOOL_GRAPHS=1 FF_GRAPH_STEPS=$k runs) ms"
  done
done
} 2>&1 | tee gpurun_out/graph/ab.txt
