# Run ON THE GPU BOX: bench.py --gpus N the way the driver launches it, with the N ranks sharing the ONE device over gloo
# (--rehearse-on-one-device: the N-rank code path, NOT a measurement) -> gpurun_out/r06r/
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06r; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for N in 2 8; do
  W=$(( N == 2 ? 128 : 16 ))
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) bench.py --gpus $N --rehearse-on-one-device --wireframes-per-gpu $W --steps 1 --warmup 1 > $O/bench_line_rehearsal_${N}ranks_one_device.json 2> $O/rehearsal_${N}.err
  echo "N=$N rc=$? $(cut -c1-300 $O/bench_line_rehearsal_${N}ranks_one_device.json)"
done
