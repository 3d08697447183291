# Run ON THE GPU BOX: the 2 x fp16 split products -- op tests, rate table, goldens in that form, config B / C128 lines of both kinds.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06c; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "x3 or x2h or split_weight" > $O/pytest_ops.log 2>&1; tail -4 $O/pytest_ops.log
timeout 600 python tools/bench_split_kinds.py > $O/gemm_split_kinds.txt 2>&1; cat $O/gemm_split_kinds.txt
rm -f $O/parity_margins_split.txt
FF_PARITY_MARGINS=$PWD/$O/parity_margins_split.txt timeout 1500 python -m pytest tests/test_parity_golden.py -m gpu -q -x -k "split_projections" > $O/pytest_split_goldens.log 2>&1; tail -4 $O/pytest_split_goldens.log
grep fp16x2 $O/parity_margins_split.txt | sort -k6 -n -r | head -8
for kind in bf16x3 fp16x2 bf16x3 fp16x2; do
  echo "== $kind"
  timeout 600 python bench.py --no-cpu-baseline --no-roofline --steps 6 --warmup 2 --split-kind $kind --other-list C128,E32 2>/dev/null > $O/bench_$kind.json
  python - $O/bench_$kind.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
x=d.get("package_default", d.get("bf16x3_projections", {}))
print("B f32 %.2f ms | B split %.2f ms %.0f edges/s" % (d["ms_per_step"], x.get("ms_per_step",-1), x.get("value",-1)))
for k,v in (d.get("other_configs") or {}).items():
    print(k, v)
PY
done
