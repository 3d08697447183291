# Run ON THE GPU BOX: (1) thresholds of the 2 x fp16 split products at config B (x3_min_rows, the 1024- / 512-column multipliers,
# LayerNorm first / in the epilogue), (2) the K | V pre-touch experiment on the f32 headline (FF_KV_TOUCH 0 / 1 / 2).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06d; mkdir -p $O
ms() { python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%.2f' % d['ms_per_step'])"; }
runx() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs --steps 8 --warmup 2 "$@" 2>/dev/null | ms; }
{
echo "# config B, split products (fp16x2) from x3_min_rows rows on; ms per wireframe, 8 passes each"
for r in 1024 512 768 1024 1280; do echo "x3_min_rows $r : $(runx --x3-min-rows $r)"; done
for n in "7 11" "6 9" "5 8" "5 7" "4 6" "6 11" "7 9"; do set -- $n; echo "min_rows 1024, need N1024 $1/4 N512 $2/4 : $(FF_X3_NEED_N1024=$1 FF_X3_NEED_N512=$2 runx --x3-min-rows 1024)"; done
for n in "6 9" "5 8" "4 6"; do set -- $n; echo "min_rows 768,  need N1024 $1/4 N512 $2/4 : $(FF_X3_NEED_N1024=$1 FF_X3_NEED_N512=$2 runx --x3-min-rows 768)"; done
echo "LayerNorm first (MODE 1), min_rows 1024 : $(FF_BENCH_LN_EPILOGUE=0 runx --x3-min-rows 1024)  / in the epilogue: $(runx --x3-min-rows 1024)"
echo "bf16x3 for reference, min_rows 1024 : $(runx --x3-min-rows 1024 --split-kind bf16x3)"
} > $O/fp16x2_thresholds.txt 2>&1
cat $O/fp16x2_thresholds.txt
{
echo "# config B f32 headline, K | V pre-touch from a side stream (FF_KV_TOUCH: 0 off, 1 beside cross-q, 2 beside the attention launch)"
for i in 1 2 3; do for k in 0 1 2; do echo "FF_KV_TOUCH=$k : $(FF_KV_TOUCH=$k runx)"; done; done
} > $O/kv_touch_ab.txt 2>&1
cat $O/kv_touch_ab.txt
timeout 900 python -m pytest tests/test_parity_golden.py -m gpu -q -x -k "tuning_table or golden_parity[par_full_B256" > $O/pytest_knobs.log 2>&1; tail -3 $O/pytest_knobs.log
FF_KV_TOUCH=1 timeout 900 python -m pytest tests/test_parity_golden.py -m gpu -q -x -k "test_golden_parity and (B256 or par_small)" > $O/pytest_touch.log 2>&1; tail -3 $O/pytest_touch.log
