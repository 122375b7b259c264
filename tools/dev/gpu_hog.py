"""dev: keeps every CU busy with long kernels for a while (robustness check of the level engine's bounded waits)."""
import sys, time, torch
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16); b = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(20): c = a @ b
    torch.cuda.synchronize(); n += 20
print("hog: %d matmuls in %.1f s" % (n, time.time() - t0))
