# Run ON THE GPU BOX (via gpurun): same-box A/B of flow launches (FF_FLOW) off / on for config B (+ optional extra bench args).
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/flow_ab
for c in 0 1 0 1; do
  timeout 600 python bench.py --flow $c --no-cpu-baseline --no-x3-line --no-other-configs --steps 8 --warmup 2 "$@" 2> gpurun_out/flow_ab/err_$c.txt | tail -1 > gpurun_out/flow_ab/line_$c.json
  python - $c <<'PY'
import json, sys
c = sys.argv[1]
d = json.loads(open("gpurun_out/flow_ab/line_%s.json" % c).read())
print("flow=%s  %.2f ms/step  launches/step %s  gemm %.2f ms (%s launches, %.1f TF/s)  attn %.2f  ln %.2f" % (
    c, d["ms_per_step"], sum(d.get("kernel_launches_per_step", {}).values()), d["kernel_time_ms_per_step"]["gemm_f32_kernels"],
    d["kernel_launches_per_step"]["gemm_f32_kernels"], d["roofline"]["achieved"], d["kernel_time_ms_per_step"]["attention_kernels"],
    d["kernel_time_ms_per_step"]["layernorm_kernel"]))
PY
done
