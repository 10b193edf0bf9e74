import torch, time
W, d = 10000, 1024
h = torch.empty((W, d), dtype=torch.float32, pin_memory=True).normal_()
g = torch.empty((W, d), dtype=torch.float32, device="cuda")
for _ in range(3): g.copy_(h, non_blocking=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): g.copy_(h, non_blocking=True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("H2D of one window (%.1f MB, pinned): %.3f ms = %.1f GB/s" % (W * d * 4 / 1e6, ms, W * d * 4 / ms / 1e6))
