"""Diagnostics: one kernel in a loop for a stated time (for power / clock / throttle sampling beside it: tools/power_probe.sh).
usage: python tools/loop_kernel.py fwd|bwd_data|wgrad|matmul SECONDS [c2|c3|e2] [masked|maskonly|colonly]  ->  prints launches, ms per launch, TFLOP/s
(c3 = the critic's 8x8x400 -> 4x4x800 conv on 1536 images, the default; c2 = 16x16x200 -> 8x8x400; `masked`: backward-data with the
lrelu derivative mask and the bias-gradient column partials of the training step's launch)"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = importlib.import_module('3dgan_amd.kernels')


def main(what, seconds, layer='c3', masked=False):
    dev = torch.device('cuda:0')
    if what == 'matmul':
        a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
        b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
        fn, fl = (lambda: torch.matmul(a, b)), 2.0 * 8192 ** 3
    else:
        n, h, w, cin, cout, k, s = {'c3': (1536, 8, 8, 400, 800, 5, 2), 'c2': (1536, 16, 16, 200, 400, 5, 2),    # the critic's c3 / c2 on [x | g | x_hat]
                                    'e2': (64, 128, 128, 64, 128, 4, 2)}[layer]                                   # pix2pix e2 (its backward-data: 64 columns, 8 K steps per class)
        big, small = K.Act(n, h, w, cin, K.BF16, dev), K.Act(n, h // 2, w // 2, cout, K.BF16, dev)
        big.buf.copy_(torch.randn_like(big.buf.float()).to(big.buf.dtype))
        small.buf.copy_(torch.randn_like(small.buf.float()).to(small.buf.dtype))
        conv = K.Conv(big, small, k, k, s, 1, 1)
        conv.pack(torch.randn(k, k, cin, cout, device=dev) * 0.05)
        dw = torch.zeros(k, k, cin, cout, device=dev)
        out = big.like()
        ws = K.Workspace(dev)

        def bwd_masked():
            if masked == 'maskonly':
                epi = K.epilogue(mask_mode=K.MASK_LRELU, leak=0.2, mask_src=big.ptr())
            elif masked == 'colonly':
                epi = K.colsum_epilogue(ws, n * h * w, cin, K.COL_SUM)
            else:
                epi = K.colsum_epilogue(ws, n * h * w, cin, K.COL_SUM, mask_mode=K.MASK_LRELU, leak=0.2, mask_src=big.ptr())
            conv.bwd_data(small.ptr(), out.ptr(), n, epi)
        fn = {'fwd': lambda: conv.fwd(big.ptr(), small.ptr(), n), 'bwd_data': bwd_masked if masked else (lambda: conv.bwd_data(small.ptr(), out.ptr(), n)),
              'wgrad': lambda: conv.bwd_filter(big.ptr(), small.ptr(), dw, n)}[what]
        fl = conv.flops(n)
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0, n_l = time.time(), 0
    while time.time() - t0 < seconds:
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        n_l += 200
    dt = time.time() - t0
    print('%s %s%s: %d launches in %.1f s, %.4f ms per launch, %.0f TFLOP/s' % (what, layer, ' ' + masked if masked else '', n_l, dt, dt / n_l * 1e3, fl * n_l / dt / 1e12), flush=True)


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else 'c3', sys.argv[4] if len(sys.argv) > 4 else False)
