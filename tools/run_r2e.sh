set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2e
timeout 600 python tools/time_seq2seq.py > gpurun_out/r2e/seq2seq.txt 2>&1; cat gpurun_out/r2e/seq2seq.txt
bash tools/collect_profiles.sh r02 > gpurun_out/r2e/collect.log 2>&1; tail -3 gpurun_out/r2e/collect.log
head -12 gpurun_out/profiles_r02/kernel_stats.md
timeout 900 python bench.py --config E --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2e/bench_E32.json 2> gpurun_out/r2e/bench_E32.err; python -c "
import json;d=json.load(open('gpurun_out/r2e/bench_E32.json'));print('E32', d['ms_per_step'], d['value'], d['wireframes_per_s'], d.get('bf16x3_projections',{}).get('ms_per_step'), d['roofline']['frac'], d['path_roofline'], d['sequence_rows'])"
