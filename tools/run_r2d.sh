set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2d
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2d/pytest.log
tail -12 gpurun_out/r2d/pytest.log
timeout 900 python bench.py > gpurun_out/r2d/bench_B.json 2> gpurun_out/r2d/bench_B.err; python -c "
import json;d=json.load(open('gpurun_out/r2d/bench_B.json'));print(d['ms_per_step'], d['value'], d.get('bf16x3_projections'), d['kernel_time_ms_per_step'], d['cpu_baseline'])"
timeout 900 python bench.py --wireframes-per-gpu 16 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2d/bench_C16.json 2> gpurun_out/r2d/bench_C16.err; python -c "
import json;d=json.load(open('gpurun_out/r2d/bench_C16.json'));print('C16', d['ms_per_step']/16, d['value'], d.get('bf16x3_projections'), d['roofline']['frac'], d['path_roofline'])"
