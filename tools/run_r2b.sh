set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2b
timeout 600 python tools/bench_gemm_ln.py 16,36 > gpurun_out/r2b/gemm_ln.txt 2>&1; cat gpurun_out/r2b/gemm_ln.txt
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-fuse-ln > gpurun_out/r2b/bench_B_nofuse.json 2> gpurun_out/r2b/bench_B.err; python -c "import json;d=json.load(open('gpurun_out/r2b/bench_B_nofuse.json'));print('unfused', d['ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r2b/bench_B.json 2> gpurun_out/r2b/bench_B.err;  python -c "import json;d=json.load(open('gpurun_out/r2b/bench_B.json'));print('fused  ', d['ms_per_step'])"
done
timeout 600 python bench.py --no-cpu-baseline --wireframes-per-gpu 16 --steps 2 --warmup 1 --no-roofline --no-fuse-ln > gpurun_out/r2b/bench_C16_nofuse.json 2> gpurun_out/r2b/bench_B.err; python -c "import json;d=json.load(open('gpurun_out/r2b/bench_C16_nofuse.json'));print('C16 unfused', d['ms_per_step']/16)"
timeout 600 python bench.py --no-cpu-baseline --wireframes-per-gpu 16 --steps 2 --warmup 1 --no-roofline > gpurun_out/r2b/bench_C16.json 2> gpurun_out/r2b/bench_B.err; python -c "import json;d=json.load(open('gpurun_out/r2b/bench_C16.json'));print('C16 fused', d['ms_per_step']/16)"
