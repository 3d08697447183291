# Run ON THE GPU BOX: the whole GPU suite on the default library (+ parity margins), the experimental launch forms' parity tests on
# libfaceformer_hip_exp.so, then the default bench line.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/parity_margins.txt
FF_PARITY_MARGINS=$PWD/gpurun_out/r04/parity_margins.txt timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/r04/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/pytest_gpu.log
tail -5 gpurun_out/r04/pytest_gpu.log
FF_HIP_LIB=$PWD/faceformer_amd/hip/libfaceformer_hip_exp.so timeout 2400 python -m pytest tests/test_parity_golden.py -m gpu -q -x -k "chain or graph or flow or engine_options" > gpurun_out/r04/pytest_gpu_experimental.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/pytest_gpu_experimental.log
tail -4 gpurun_out/r04/pytest_gpu_experimental.log
timeout 1500 python bench.py "$@" > gpurun_out/r04/bench_r04_B.json 2> gpurun_out/r04/bench_r04_B.err; tail -c 400 gpurun_out/r04/bench_r04_B.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04/bench_r04_B.json").read().strip().splitlines()[-1])
print("B f32", d["ms_per_step"], d["value"], "gemm frac", d["roofline"]["frac"], "path", d["path_roofline"]["frac_of_f32_mfma_peak"])
x=d["bf16x3_projections"]; print("B x3", x["ms_per_step"], x["value"], x["roofline"]["achieved"], x["roofline"]["frac"])
for k,v in d["other_configs"].items():
    print(k, round(v["ms_per_step"],1), round(v["value"]), "x3:", round((v.get("bf16x3_projections") or {}).get("value") or 0))
print("cpu", (d.get("cpu_baseline") or {}).get("value"), d.get("speedup_vs_cpu"))
PY
