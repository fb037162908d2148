#!/usr/bin/env python3
"""Does RCCL accept two ranks on ONE device?  (NCCL refuses: 'Duplicate GPU detected'.)  If it does, the native
multi-GPU driver can be exercised over real RCCL channels on a single-GPU box."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def w(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        t = torch.full((1024,), float(rank), device="cuda")
        if rank == 0:
            dist.send(t, 1)
        else:
            dist.recv(t, 0)
        torch.cuda.synchronize()
        print("rank", rank, "ok", float(t[0]), flush=True)
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        print("rank", rank, "FAILED:", repr(e)[:300], flush=True)
        sys.exit(3)


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=w, args=(r, 2, 29611)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(100)
        if p.is_alive():
            p.kill()
    print("exit codes", [p.exitcode for p in ps])
