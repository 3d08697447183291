#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the round-4 profile set -> gpurun_out/profiles_r04/ (copied to profiles/r04/ afterwards).
#   kernel trace + the five PMC passes of the default bench command (f32 headline) and of the package-default form
#   (--x3-min-rows 1024: kernel trace + SQ / FETCH / WRITE passes), per-step breakdowns, the projection-kernel tables, the
#   LayerNorm-folded forms, zero-filled operands, attention sweep, seq2seq timings, the bf16 MFMA pattern rates.
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_r04
mkdir -p $OUT
bash tools/collect_profiles.sh r04 > $OUT/collect.log 2>&1
# the package-default form: its own trace and counter passes
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
  CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs --x3-min-rows 1024"
  echo "command: $CMD" > $OUT/command_x3.txt
  rm -rf /tmp/prof_x3 && rocprofv3 --kernel-trace --stats -d /tmp/prof_x3 -o t -- $CMD > $OUT/trace_x3.log 2>&1
  python tools/rocpd_stats.py /tmp/prof_x3/t_results.db $OUT/kernel_stats_x3.md > /dev/null
  python tools/step_breakdown.py /tmp/prof_x3/t_results.db 8,20,33,36 > $OUT/steps_x3.txt 2>&1
  for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
    set -- $pass; name=$1; shift
    rm -rf /tmp/prof_pmc && rocprofv3 --pmc "$@" -d /tmp/prof_pmc -o p -- $CMD > /dev/null 2>&1
    python tools/pmc_per_kernel.py /tmp/prof_pmc/p_results.db $OUT/pmc_${name}_x3.md > /dev/null
  done
  rm -rf /tmp/prof_pmc /tmp/prof_x3 )
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && rm -rf /tmp/prof_trace2 &&
  rocprofv3 --kernel-trace --stats -d /tmp/prof_trace2 -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs > /dev/null 2>&1 &&
  python tools/step_breakdown.py /tmp/prof_trace2/t_results.db 4,8,20,36 > $OUT/steps.txt 2>&1; rm -rf /tmp/prof_trace2 )
python tools/make_traffic_json.py $OUT > /dev/null 2>&1
timeout 900 python tools/bench_gemm.py --ts 8,16,24,36,64,128 --tiles 3,7,11 --x3-variants > $OUT/gemm_tiles_vs_vendor.txt 2>&1
timeout 900 python tools/bench_gemm.py --ts 36,128 --tiles 7,11 --data zeros > $OUT/gemm_zero_operands.txt 2>&1
timeout 600 python tools/bench_gemm_x3_ln.py --ms 2048,4096,6144,9216,32768 > $OUT/gemm_ln_forms.txt 2>&1
timeout 600 python tools/time_seq2seq.py > $OUT/seq2seq.txt 2>&1
timeout 600 python tools/attn_sweep.py > $OUT/attention_sweep.txt 2>&1
for u in mfma_bf16 x3v2; do
  hipcc --offload-arch=gfx950 -O3 tools/ubench/$u.hip -o /tmp/$u 2>/dev/null
done
timeout 120 /tmp/mfma_bf16 > $OUT/ubench_mfma_bf16.txt 2>&1
timeout 200 /tmp/x3v2 > $OUT/x3v2_probe_random.txt 2>&1
timeout 200 /tmp/x3v2 --zeros > $OUT/x3v2_probe_zeros.txt 2>&1
ls -la $OUT
