"""Does torch's gloo backend move CUDA (HIP) tensors on this image?  Two ranks on cuda:0 (run on the GPU box)."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp


def work(rank, world, port):
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    for name, fn in (
        ("all_reduce int64", lambda: dist.all_reduce(torch.full((4,), rank + 1, dtype=torch.int64, device=dev))),
        ("all_gather_into_tensor int32", lambda: dist.all_gather_into_tensor(torch.empty(8, dtype=torch.int32, device=dev),
                                                                            torch.full((4,), rank, dtype=torch.int32, device=dev))),
        ("all_gather_into_tensor uint8", lambda: dist.all_gather_into_tensor(torch.empty(8, dtype=torch.uint8, device=dev),
                                                                            torch.full((4,), rank, dtype=torch.uint8, device=dev))),
        ("barrier", lambda: dist.barrier()),
    ):
        try:
            fn()
            torch.cuda.synchronize()
            print("rank %d: %s ok" % (rank, name))
        except Exception as e:   # noqa: BLE001
            print("rank %d: %s FAILED: %s" % (rank, name, str(e)[:200]))
        sys.stdout.flush()
    dist.destroy_process_group()


if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=work, args=(r, 2, port)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(120)
        print("exit", p.exitcode)
