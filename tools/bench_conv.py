"""Micro-benchmark of one conv GEMM shape through the C ABI (diagnostics; not part of the product path).
usage: python tools/bench_conv.py [n h w cin cout k stride] [--dtype bf16|f32]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = importlib.import_module('3dgan_amd.kernels')


def run(n, h, w, cin, cout, k, s, dtype, reps=20):
    dev = torch.device('cuda:0')
    oh, ow = -(-h // s), -(-w // s)
    pt = max((oh - 1) * s + k - h, 0) // 2
    big, small = K.Act(n, h, w, cin, dtype, dev), K.Act(n, oh, ow, cout, dtype, dev)
    big.buf.copy_(torch.randn_like(big.buf.float()).to(big.buf.dtype))
    small.buf.copy_(torch.randn_like(small.buf.float()).to(small.buf.dtype))
    conv = K.Conv(big, small, k, k, s, pt, pt)
    conv.pack(torch.randn(k, k, cin, cout, device=dev) * 0.05)
    dw = torch.zeros(k, k, cin, cout, device=dev)
    out = big.like()
    fl = conv.flops(n)
    res = {}
    for name, fn in (('fwd', lambda: conv.fwd(big.ptr(), small.ptr(), n)),
                     ('bwd_data', lambda: conv.bwd_data(small.ptr(), out.ptr(), n)),
                     ('bwd_filter', lambda: conv.bwd_filter(big.ptr(), small.ptr(), dw, n))):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        res[name] = (ms, fl / ms / 1e9)
    return res


if __name__ == '__main__':
    dtype = K.F32 if '--dtype' in sys.argv and sys.argv[sys.argv.index('--dtype') + 1] == 'f32' else K.BF16
    nums = [int(a) for a in sys.argv[1:] if a.lstrip('-').isdigit()]
    shapes = [tuple(nums)] if len(nums) == 7 else [
        (1536, 16, 16, 200, 400, 5, 2),      # D c2 on [x|g|x_hat]
        (1536, 8, 8, 400, 800, 5, 2),        # D c3
        (1536, 32, 32, 3, 200, 5, 2),        # D c1 (thin)
        (512, 4, 4, 400, 800, 5, 2),         # G dc1 (as conv between 4x4x400 and 2x2x800)
        (512, 8, 8, 200, 400, 5, 2),         # G dc2
        (512, 16, 16, 104, 200, 5, 2),       # G dc3
        (512, 32, 32, 3, 100, 5, 2),         # G dc4 (thin)
    ]
    for sh in shapes:
        r = run(*sh, dtype)
        print(sh, ' '.join('%s %.3fms %.0fTF' % (k, v[0], v[1]) for k, v in r.items()), flush=True)
