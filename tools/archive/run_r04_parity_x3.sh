# Run ON THE GPU BOX: parity goldens in the bf16-split forms + op tests of the split kernel + a short config-B bench line
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/parity_margins_x3.txt
FF_PARITY_MARGINS=$PWD/gpurun_out/r04/parity_margins_x3.txt timeout 2400 python -m pytest tests/test_parity_golden.py tests/test_hip_ops.py -m gpu -q -x -k "bf16_split or x3 or test_golden_parity[ or fresh_inputs" > gpurun_out/r04/parity_x3.log 2>&1
tail -6 gpurun_out/r04/parity_x3.log
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/r04/bench_B_x3.json 2> gpurun_out/r04/bench_B_x3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_B_x3.json').read().strip().splitlines()[-1])
print('f32 ms', d['ms_per_step'], 'x3 ms', d['bf16x3_projections']['ms_per_step'])
print(d.get('kernel_time_ms_per_step'))
PY
