# Run ON THE GPU BOX: round 6, first session -- op tests of the swizzled statistics patch, A/B of write-through result stores
# (build_ub/lib_wt.so, -DFF_ST_WT) against round 5's library (build_ub/lib_r05.so) and the tree, per-shape GEMM table.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06a; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm or linear or attention" > $O/pytest_ops.log 2>&1; tail -2 $O/pytest_ops.log
FF_HIP_LIB=build_ub/lib_wt.so timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm or linear or attention" > $O/pytest_ops_wt.log 2>&1; tail -2 $O/pytest_ops_wt.log
runb() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs --steps 8 --warmup 2 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%.2f' % d['ms_per_step'])"; }
runs() { FF_SEQ_ONLY_A=1 timeout 300 python tools/time_seq2seq.py 2>&1 | grep seq2seq | awk '{print $5}'; }
{
for i in 1 2 3; do
  echo "r05     : B $(FF_HIP_LIB=build_ub/lib_r05.so runb) ms"
  echo "in-tree : B $(runb) ms"
  echo "wt      : B $(FF_HIP_LIB=build_ub/lib_wt.so runb) ms"
done
echo "seq2seq A: r05 $(FF_HIP_LIB=build_ub/lib_r05.so runs)  in-tree $(runs)  wt $(FF_HIP_LIB=build_ub/lib_wt.so runs)"
} > $O/wt_ab.txt 2>&1
cat $O/wt_ab.txt
# per-shape table: trace of the tree + the isolated loops
rm -rf /tmp/prof_trace
rocprofv3 --kernel-trace --stats -d /tmp/prof_trace -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs > $O/trace.log 2>&1
python tools/rocpd_stats.py /tmp/prof_trace/t_results.db $O/kernel_stats.md > /dev/null
python tools/step_breakdown.py /tmp/prof_trace/t_results.db 1,2,33,36 > $O/steps.txt 2>&1
timeout 600 python tools/bench_gemm_ln.py 9,12,16,20,24,25,28,32,33,36 > $O/gemm_ln_isolated.txt 2>&1
python tools/gemm_by_shape.py /tmp/prof_trace/t_results.db 9,12,16,20,24,25,28,32,33,36 $O/gemm_ln_isolated.txt > $O/gemm_by_shape_in_decode.txt 2>&1
head -50 $O/steps.txt
cat $O/gemm_by_shape_in_decode.txt
# same trace with the write-through build
rm -rf /tmp/prof_trace
FF_HIP_LIB=build_ub/lib_wt.so rocprofv3 --kernel-trace --stats -d /tmp/prof_trace -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs > $O/trace_wt.log 2>&1
python tools/rocpd_stats.py /tmp/prof_trace/t_results.db $O/kernel_stats_wt.md > /dev/null
python tools/step_breakdown.py /tmp/prof_trace/t_results.db > $O/steps_wt.txt 2>&1
head -22 $O/kernel_stats_wt.md
# LDS bank conflicts of the LayerNorm-consuming LDS-DMA kernel (was 0.40 of its LDS cycles)
rm -rf /tmp/prof_pmc
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 -d /tmp/prof_pmc -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs > $O/pmc_lds.log 2>&1
python tools/pmc_per_kernel.py /tmp/prof_pmc/p_results.db $O/pmc_lds.md > /dev/null
head -12 $O/pmc_lds.md
