#!/usr/bin/env python3
"""Library DGEMM rate on this GPU for the shape of the Cholesky trailing update (C[n,n] -= A[n,K] A[n,K]^T, K = 512)
and for a square product: the yardstick for k_syrk_mfma (torch.mm -> rocBLAS/hipBLASLt fp64)."""
import time, torch
dev = torch.device("cuda:0")
for (m, n, k) in ((16384, 16384, 512), (32768, 32768, 512), (8192, 8192, 8192)):
    A = torch.randn(m, k, dtype=torch.float64, device=dev)
    B = torch.randn(n, k, dtype=torch.float64, device=dev)
    C = torch.zeros(m, n, dtype=torch.float64, device=dev)
    for _ in range(2):
        torch.addmm(C, A, B.t(), beta=1.0, alpha=-1.0, out=C)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        torch.addmm(C, A, B.t(), beta=1.0, alpha=-1.0, out=C)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / reps
    print("dgemm %6d x %6d x %5d: %8.2f ms  %6.2f TFLOP/s" % (m, n, k, t * 1e3, 2.0 * m * n * k / t / 1e12), flush=True)
    del A, B, C
