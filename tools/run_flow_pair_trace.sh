# Run ON THE GPU BOX (via gpurun): the review's kill criterion for fused dependent projections, measured on the path --
# kernel-by-kernel durations of decode steps 20 (5120 rows) and 36 (9216 rows) of config B with one launch per projection
# (LayerNorm-folded forms at every size) and with PAIRS of dependent projections in one flow launch (FF_FLOW, FF_FLOW_MAX_OPS=2:
# out-proj -> cross-q is such a pair).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/flowpair
for mode in launches pairs; do
  rm -rf /tmp/prof_fp
  if [ $mode = pairs ]; then export FF_FLOW_MAX_OPS=2; extra="--flow 1 --flow-min-rows 1"; else unset FF_FLOW_MAX_OPS; extra="--flow 0 --ln-fuse-max-rows 1000000"; fi
  rocprofv3 --kernel-trace -d /tmp/prof_fp -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs $extra > gpurun_out/flowpair/trace_$mode.log 2>&1
  python tools/step_breakdown.py /tmp/prof_fp/t_results.db 20,36 > gpurun_out/flowpair/steps_$mode.txt 2>&1
done
{
for t in 20 36; do
  for mode in launches pairs; do
    echo "==== step $t, $mode: first decoder layers"
    awk -v t=$t '$0 ~ "^---- step "t"$" {p=1; n=0; next} /^---- step/ {p=0} p && n < 20 {print; n++}' gpurun_out/flowpair/steps_$mode.txt
    grep -E "^ *$t " gpurun_out/flowpair/steps_$mode.txt | head -1
  done
done
} | tee gpurun_out/flowpair/flow_pair_trace.txt
