#!/usr/bin/env python3
"""dev: all_to_all_single / all_gather_into_tensor through RCCL with ONE rank: how much of a large buffer arrives."""
import os, sys, socket
os.environ.update({"MASTER_ADDR": "127.0.0.1"})
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s.getsockname()[1])
import torch
import torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for mb in (64, 256, 512, 600, 679, 680, 700, 1024, 1358, 2047, 2049, 3000):
    n = mb * (1 << 20) // 16
    send = torch.arange(n * 4, dtype=torch.int32, device=dev).reshape(n, 4)
    recv = torch.full((n, 4), -1, dtype=torch.int32, device=dev)
    dist.all_to_all_single(recv, send, [n], [n])
    torch.cuda.synchronize()
    bad = torch.nonzero((recv != send).any(dim=1)).flatten()
    g = torch.full((n, 4), -1, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(g, send)
    torch.cuda.synchronize()
    badg = torch.nonzero((g != send).any(dim=1)).flatten()
    print("%5d MiB: all_to_all rows wrong %d (first %s = %.1f MiB)   all_gather rows wrong %d" % (
        mb, bad.numel(), bad[:1].tolist(), (bad[0].item() * 16 / 2**20) if bad.numel() else -1, badg.numel()), flush=True)
    del send, recv, g
dist.destroy_process_group()
