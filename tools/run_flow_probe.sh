# Run ON THE GPU BOX: timing probes of the flow launches (config B): baseline, flow, flow without dependencies (wrong results,
# timing only), one operator per flow launch.
set -u
cd "$GRAFT_REPO_ROOT"
run() { timeout 600 python bench.py --no-cpu-baseline --no-x3-line --no-other-configs --steps 8 --warmup 2 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%.2f ms/step  gemm %.2f ms (%d launches)  attn %.2f  ln %.2f' % (d['ms_per_step'], d['kernel_time_ms_per_step']['gemm_f32_kernels'], d['kernel_launches_per_step']['gemm_f32_kernels'], d['kernel_time_ms_per_step']['attention_kernels'], d['kernel_time_ms_per_step']['layernorm_kernel']))"; }
echo "baseline            : $(run --flow 0)"
echo "LN folded always    : $(run --flow 0 --ln-fuse-max-rows 1000000)"
echo "flow                : $(run --flow 1)"
echo "flow, no deps       : $(FF_FLOW_NODEP=1 run --flow 1)"
echo "flow, 1 op/launch   : $(FF_FLOW_MAX_OPS=1 run --flow 1)"
echo "flow from 4097 rows : $(run --flow 1 --flow-min-rows 4097)"
echo "flow up to 4096 rows: $(FF_FLOW_MAX_ROWS=4096 run --flow 1)"
