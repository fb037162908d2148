import torch, time
dev = torch.device("cuda", 0)
M = 16384
a = torch.randn(M, M, dtype=torch.float64, device=dev); b = torch.randn(M, M, dtype=torch.float64, device=dev); c = torch.zeros(M, M, dtype=torch.float64, device=dev)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(40): c.addmm_(a, b, beta=0.0, alpha=1.0)
torch.cuda.synchronize(); t = time.time() - t0
print("vendor DGEMM %.1f TFLOP/s over %.1f s" % (40 * 2.0 * M**3 / t * 1e-12, t))
