# Run ON THE GPU BOX (via gpurun): same-box A/B of chain launches (FF_CHAIN) off / on: config B headline, seq2seq configs D / A.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/chain_ab
for c in 0 1 0 1; do
  timeout 600 python bench.py --chain $c --no-cpu-baseline --no-x3-line --other-list D,A --steps 8 --warmup 2 "$@" 2> gpurun_out/chain_ab/err_$c.txt | tail -1 > gpurun_out/chain_ab/line_$c.json
  python - $c <<'PY'
import json, sys
c = sys.argv[1]
d = json.loads(open("gpurun_out/chain_ab/line_%s.json" % c).read())
o = d.get("other_configs", {})
print("chain=%s  B %.2f ms  D %.2f ms  A %.2f ms  launches/step B %s" % (
    c, d["ms_per_step"], o.get("D", {}).get("ms_per_step", -1), o.get("A", {}).get("ms_per_step", -1),
    sum(d.get("kernel_launches_per_step", {}).values())))
print("   B kernel ms:", {k: round(v, 2) for k, v in d.get("kernel_time_ms_per_step", {}).items()})
print("   A kernel ms:", {k: round(v, 2) for k, v in o.get("A", {}).get("kernel_time_ms_per_step", {}).items()},
      "launches", o.get("A", {}).get("kernel_launches_per_step"))
PY
done
