#!/usr/bin/env python3
"""Calibration: what the vendor DGEMM (rocBLAS/hipBLASLt through torch.mm, fp64) reaches on this box, for the
square case and for the rank-128 update shape C(MxM) -= A(Mx128) B(128xM) that k_update's pieces have."""

import torch

dev = torch.device("cuda", 0)


def run(M, N, K, reps=5):
    a = torch.randn(M, K, dtype=torch.float64, device=dev)
    b = torch.randn(K, N, dtype=torch.float64, device=dev)
    c = torch.randn(M, N, dtype=torch.float64, device=dev)
    for _ in range(2):
        c.addmm_(a, b, beta=1.0, alpha=-1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        c.addmm_(a, b, beta=1.0, alpha=-1.0)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / reps
    print("DGEMM M=%d N=%d K=%d: %.3f ms  %.1f TFLOP/s" % (M, N, K, t * 1e3, 2.0 * M * N * K / t * 1e-12), flush=True)


for M, N, K in [(8192, 8192, 8192), (16384, 16384, 16384), (16384, 16384, 128), (32768, 32768, 128), (32768, 32768, 2048)]:
    run(M, N, K)
