#!/bin/bash
# Run ON THE GPU BOX: layer-0 LayerNorm folded into the newest rows' q|k|v projection (statistics from the pointer launch) against
# the standalone LayerNorm launch (FF_L0_FOLD=0), alternating in one job -> gpurun_out/l0fold_ab.txt
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/l0fold_ab.txt
: > $OUT
for rep in 1 2 3; do
  for v in 1 0; do
    echo "== FF_L0_FOLD=$v rep $rep" >> $OUT
    FF_L0_FOLD=$v python bench.py --no-cpu-baseline --no-other-configs --no-roofline --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config B f32 %.3f ms   package default %.3f ms' % (d['ms_per_step'], d['bf16x3_projections']['ms_per_step']))" >> $OUT
    FF_L0_FOLD=$v FF_SEQ_REPS=5 FF_SEQ_BATCHES=1,64 python tools/time_seq2seq.py 2>&1 | grep "ms " >> $OUT
  done
done
cat $OUT
