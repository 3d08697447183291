#!/bin/bash
# Run ON THE GPU BOX: pruned last layer's q | k | v as ONE launch over all rows up to N active rows (FF_LAST_QKV_ONE_LAUNCH_ROWS)
# against k | v + newest-row q as two launches (0), alternating -> gpurun_out/lastqkv_ab.txt
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/lastqkv_ab.txt
: > $OUT
for rep in 1 2 3; do
  for v in 0 1024 4096; do
    echo "== FF_LAST_QKV_ONE_LAUNCH_ROWS=$v rep $rep" >> $OUT
    FF_LAST_QKV_ONE_LAUNCH_ROWS=$v python bench.py --no-cpu-baseline --no-other-configs --no-roofline --no-x3-line --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config B f32 %.3f ms' % d['ms_per_step'])" >> $OUT
    FF_LAST_QKV_ONE_LAUNCH_ROWS=$v FF_SEQ_REPS=5 FF_SEQ_BATCHES=1,8,64 python tools/time_seq2seq.py 2>&1 | grep "ms " >> $OUT
  done
done
cat $OUT
