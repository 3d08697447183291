# Run ON THE GPU BOX (via gpurun): per-step wall time of config B as one launch chain vs two sequence groups on two streams.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/split
for g in 1 2; do
  rm -rf /tmp/prof_split
  extra=""; [ $g = 2 ] && extra="--chunk-seqs 128 --streams 2"
  FF_NO_GRAPH=1 rocprofv3 --kernel-trace -d /tmp/prof_split -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs $extra > gpurun_out/split/trace_$g.log 2>&1
  python tools/step_walls.py /tmp/prof_split/t_results.db $g > gpurun_out/split/walls_$g.txt 2>&1
done
paste gpurun_out/split/walls_1.txt gpurun_out/split/walls_2.txt
