#!/bin/bash
# Run ON THE GPU BOX: decoder.norm + project + pointer dot products of a one-wireframe micro-batch as ONE launch
# (logits = LN(x) (memory W')^T + memory b'; FF_POINTER_FOLD=1, default) against project and the pointer GEMM as two launches (0),
# alternating -> gpurun_out/pointer_fold_ab.txt
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pointer_fold_ab.txt
: > $OUT
for rep in 1 2 3; do
  for v in 1 0; do
    echo "== FF_POINTER_FOLD=$v rep $rep" >> $OUT
    FF_POINTER_FOLD=$v python bench.py --no-cpu-baseline --no-other-configs --no-roofline --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config B f32 %.3f ms   package default %.3f ms' % (d['ms_per_step'], d['bf16x3_projections']['ms_per_step']))" >> $OUT
    FF_POINTER_FOLD=$v FF_SEQ_REPS=5 FF_SEQ_BATCHES=1,64 python tools/time_seq2seq.py 2>&1 | grep "ms " >> $OUT
  done
done
cat $OUT
