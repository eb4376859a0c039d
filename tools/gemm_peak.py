"""Best fp16 GEMM rate torch/hipBLASLt reaches on this box (context for the tower's MFMA fraction)."""
import torch
best = 0
for n in (4096, 8192, 16384):
    a = torch.randn(n, n, device='cuda', dtype=torch.float16); b = torch.randn(n, n, device='cuda', dtype=torch.float16)
    for _ in range(3): a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): a @ b
    e1.record(); torch.cuda.synchronize()
    tf = 2 * n ** 3 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    print('fp16 GEMM %d^3: %.0f TFLOP/s' % (n, tf)); best = max(best, tf)
print('best %.0f TFLOP/s' % best)
