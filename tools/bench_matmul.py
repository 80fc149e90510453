"""Diagnostics: library GEMM rate at the conv GEMM shapes (what hipBLASLt reaches on this box)."""
import sys
import torch
dev = torch.device('cuda:0')
shapes = [(98304, 5000, 400), (24576, 10000, 800), (98304, 5120, 416), (24576, 10240, 832), (8192, 8192, 8192)]
if len(sys.argv) == 4:                      # one shape: m k n
    shapes = [tuple(int(v) for v in sys.argv[1:4])]
for (m, k, n) in shapes:
    a = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    b = torch.randn(k, n, device=dev, dtype=torch.bfloat16)
    bt = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
    for name, fn in (('NN', lambda: a @ b), ('NT', lambda: a @ bt.t())):
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        print(m, k, n, name, '%.3f ms %.0f TF' % (ms, 2.0 * m * k * n / ms / 1e9), flush=True)
