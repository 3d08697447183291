set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2f
bash tools/collect_profiles.sh r02 > gpurun_out/r2f/collect.log 2>&1; tail -2 gpurun_out/r2f/collect.log
python tools/step_breakdown.py /tmp/prof_trace/t_results.db 2,8,16,24,32,33,36 > gpurun_out/profiles_r02/steps.txt 2>&1
timeout 900 python bench.py > gpurun_out/r2f/bench_r02_B.json 2> gpurun_out/r2f/err.txt; tail -c 300 gpurun_out/r2f/bench_r02_B.json
timeout 900 python bench.py --wireframes-per-gpu 128 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2f/bench_r02_C128.json 2>> gpurun_out/r2f/err.txt; python -c "
import json;d=json.load(open('gpurun_out/r2f/bench_r02_C128.json'));print('C128', d['ms_per_step']/128, d['value'], d.get('bf16x3_projections',{}).get('value'), d['roofline']['frac'], d['path_roofline']['frac_of_f32_mfma_peak'])"
timeout 900 python bench.py --config E --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2f/bench_r02_E32.json 2>> gpurun_out/r2f/err.txt; python -c "
import json;d=json.load(open('gpurun_out/r2f/bench_r02_E32.json'));print('E32', d['ms_per_step'], d['value'], d.get('bf16x3_projections',{}).get('value'))"
timeout 900 python bench.py --config E --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-x3-line --no-dedup > gpurun_out/r2f/bench_r02_E32_nodedup.json 2>> gpurun_out/r2f/err.txt; python -c "
import json;d=json.load(open('gpurun_out/r2f/bench_r02_E32_nodedup.json'));print('E32 nodedup', d['ms_per_step'], d['value'])"
