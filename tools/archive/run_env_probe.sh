# Run ON THE GPU BOX (via gpurun): HIP runtime knobs that change the cost of a dependent launch; config B and config A (one wireframe).
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/envprobe
runb() { timeout 300 python bench.py ${BARGS:-} --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%.2f' % d['ms_per_step'])"; }
runs() { FF_SEQ_ONLY_A=1 timeout 600 python tools/time_seq2seq.py 2>&1 | grep seq2seq | awk '{print $5}' | tr '\n' ' '; }
try() { echo "$*: B $(env "$@" runb_) ms"; }
{
for i in 1 2; do
  echo "plain launches: B $(runb) ms  seq2seq $(runs) ms"
  for kv in HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=0 AMD_OPT_FLUSH=0 AMD_OPT_FLUSH=1 GPU_MAX_HW_QUEUES=1 ROC_SYSTEM_SCOPE_SIGNAL=0 DEBUG_HIP_KERNARG_COPY_OPT=0 ROC_USE_FGS_KERNARG=0 ROC_USE_FGS_KERNARG=1; do
    echo "plain launches, $kv: B $(export $kv; runb) ms  seq2seq $(export $kv; runs) ms"
  done
  for kv in X=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 DEBUG_HIP_GRAPH_BATCH_SIZE=1000 HIP_FORCE_DEV_KERNARG=1; do
    echo "graphs of 4 steps, $kv: B $(export BARGS="--graphs 1" FF_GRAPH_STEPS=4 $kv; runb) ms  seq2seq $(export FF_TOOL_GRAPHS=1 FF_GRAPH_STEPS=4 $kv; runs) ms"
  done
done
} 2>&1 | tee gpurun_out/envprobe/ab.txt
