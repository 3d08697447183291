#!/bin/bash
# Run ON THE GPU BOX: row limit of the unstaged small-M kernel (ff_set_gemm_tuning small_max_rows; wide outputs leave at 3/4 of it)
# re-swept with the hybrid stream-K launch in the tree -> gpurun_out/small_max_rows_ab.txt
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/small_max_rows_ab.txt
: > $OUT
for rep in 1 2 3; do
  for v in 1024 512 768 683 1366; do
    echo "== small_max_rows=$v rep $rep" >> $OUT
    python bench.py --gemm-tuning 2,2048,25,$v --no-cpu-baseline --no-other-configs --no-roofline --no-x3-line --steps 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config B f32 %.3f ms' % d['ms_per_step'])" >> $OUT
  done
done
cat $OUT
