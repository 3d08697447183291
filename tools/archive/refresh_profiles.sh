#!/bin/bash
# Run ON THE GPU BOX (via gpurun): regenerate every file of profiles/<tag>/ with the current build --
# kernel trace + PMC passes (collect_profiles.sh), per-step breakdown, the bench lines of configs B / C / E,
# the seq2seq timings, the attention sweeps, the projection-kernel tables (tile variants vs the vendor library, zero-filled
# operands, LayerNorm-fused forms), the vendor kernel names and the MFMA micro-benchmarks.  Results land in gpurun_out/profiles_<tag>/.
set -u
TAG=${1:-r03}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
bash tools/collect_profiles.sh $TAG > $OUT/collect.log 2>&1
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && rm -rf /tmp/prof_trace2 &&
  rocprofv3 --kernel-trace --stats -d /tmp/prof_trace2 -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs > /dev/null 2>&1 &&
  python tools/step_breakdown.py /tmp/prof_trace2/t_results.db 4,8,20,36 > $OUT/steps.txt 2>&1; rm -rf /tmp/prof_trace2 )
timeout 900 python bench.py > $OUT/bench_${TAG}_B.json 2> $OUT/bench_B.err
timeout 600 python bench.py --wireframes-per-gpu 128 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/bench_${TAG}_C128.json 2>> $OUT/bench_B.err
timeout 600 python bench.py --config E --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/bench_${TAG}_E32.json 2>> $OUT/bench_B.err
timeout 600 python bench.py --config E --steps 2 --warmup 1 --no-cpu-baseline --no-dedup --no-x3-line --no-other-configs > $OUT/bench_${TAG}_E32_nodedup.json 2>> $OUT/bench_B.err
timeout 600 python tools/time_seq2seq.py > $OUT/seq2seq.txt 2>&1
timeout 600 python tools/attn_sweep.py > $OUT/attention_sweep.txt 2>&1
timeout 600 python tools/attn_sweep.py --seq > $OUT/attention_sweep_seq2seq.txt 2>&1
timeout 900 python tools/bench_gemm.py --ts 16,36,64,128 --tiles 2,3,4,5,7 > $OUT/gemm_tiles_vs_vendor.txt 2>&1
timeout 900 python tools/bench_gemm.py --ts 36,128 --tiles 3,7 --data zeros > $OUT/gemm_zero_operands.txt 2>&1
timeout 600 python tools/bench_gemm_ln.py > $OUT/gemm_ln_fusion.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && rm -rf /tmp/vn &&
  rocprofv3 --kernel-trace --stats -d /tmp/vn -o v -- python tools/vendor_names.py > /dev/null 2>&1 &&
  python tools/rocpd_stats.py /tmp/vn/v_results.db $OUT/vendor_kernels.md > /dev/null 2>&1; rm -rf /tmp/vn )
for u in mfma_chain mfma_lds grid_phase launch_floor; do
  hipcc --offload-arch=gfx950 -O3 tools/ubench/$u.hip -o /tmp/$u 2>/dev/null && timeout 120 /tmp/$u > $OUT/ubench_$u.txt 2>&1
done
python tools/make_traffic_json.py $OUT > /dev/null 2>&1
# round 3: the persistent-launch experiments (opt-in FF_CHAIN / FF_FLOW): per-operator trace, same-box A/Bs
FF_CHAIN_TRACE=100 timeout 300 python tools/trace_chain.py 2> $OUT/chain_trace.txt > /dev/null
bash tools/run_chain_ab.sh > $OUT/chain_ab.txt 2>&1
bash tools/run_flow_probe.sh > $OUT/flow_probe.txt 2>&1
bash tools/run_ln_probe.sh > $OUT/ln_fold_sweep.txt 2>&1
ls -la $OUT
tail -c 1500 $OUT/bench_${TAG}_B.json
