set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2c
timeout 900 python tools/attn_sweep.py > gpurun_out/r2c/attn_sweep.txt 2>&1; cat gpurun_out/r2c/attn_sweep.txt
