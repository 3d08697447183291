set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2c
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "attention" > gpurun_out/r2c/pytest_attn.log 2>&1; tail -3 gpurun_out/r2c/pytest_attn.log
bash tools/run_trace.sh > /dev/null 2>&1
python - <<'PY'
import re
txt=open('gpurun_out/trace/steps.txt').read()
print(txt.split('---- step')[0][-1500:])
for blk in txt.split('---- step')[1:]:
    lines=blk.strip().split('\n'); t=lines[0].strip()
    agg={}
    for l in lines[1:]:
        m=re.match(r'(\S+(?:<[^>]*>)?)\s+([\d.]+) us', l.strip())
        if m:
            a=agg.setdefault(m.group(1),[0,0.0,[]]); a[0]+=1; a[1]+=float(m.group(2)); a[2].append(float(m.group(2)))
    print('step',t)
    for k,(n,s,v) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:6]:
        print('   %-34s n=%3d total %7.1f  each: %s' % (k,n,s,' '.join('%.1f'%x for x in v[:8])))
PY
