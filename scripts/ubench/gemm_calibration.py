#!/usr/bin/env python3
"""Calibration of the dense-f16 peak on THIS box: what the vendor GEMM (torch.matmul -> hipBLASLt /
rocBLAS) sustains on random and on all-zero fp16 operands, fp32 accumulate.  The screen kernel's
roofline.frac is quoted against the 2.5 PFLOP/s spec peak; this is the number a tuned library reaches
under the same power limit."""
import json
import time

import torch


def run(n, zero, iters=30):
    dev = torch.device("cuda", 0)
    a = torch.zeros((n, n), dtype=torch.float16, device=dev) if zero else \
        torch.randn((n, n), dtype=torch.float16, device=dev)
    b = torch.zeros((n, n), dtype=torch.float16, device=dev) if zero else \
        torch.randn((n, n), dtype=torch.float16, device=dev)
    for _ in range(5):
        c = a @ b
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        c = a @ b
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return 2.0 * n ** 3 / dt / 1e12, float(c[0, 0])


out = {}
for n in (4096, 8192, 16384):
    for zero in (False, True):
        tf, _ = run(n, zero)
        out["n{}_{}".format(n, "zeros" if zero else "randn")] = round(tf, 1)
# sustained: 3 s of back-to-back random GEMMs (clocks settle under the power limit)
n = 8192
a = torch.randn((n, n), dtype=torch.float16, device="cuda")
b = torch.randn((n, n), dtype=torch.float16, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
k = 0
while time.perf_counter() - t0 < 3.0:
    for _ in range(20):
        c = a @ b
    torch.cuda.synchronize()
    k += 20
out["n8192_randn_sustained_3s"] = round(2.0 * n ** 3 * k / (time.perf_counter() - t0) / 1e12, 1)
out["unit"] = "TFLOP/s (fp16 in, fp32 accumulate, torch.matmul)"
print(json.dumps(out))
