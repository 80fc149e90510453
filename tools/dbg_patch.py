"""Diagnostics: igemm_fwd_patch_kernel against igemm_fwd_dma_kernel on the same inputs, element by element."""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = importlib.import_module('3dgan_amd.kernels')
L = importlib.import_module('3dgan_amd._lib')

def run(n, h, w, cin, cout, k, s, reps=3):
    dev = torch.device('cuda:0')
    oh, ow = -(-h // s), -(-w // s)
    pt = max((oh - 1) * s + k - h, 0) // 2
    big, small = K.Act(n, h, w, cin, K.BF16, dev), K.Act(n, oh, ow, cout, K.BF16, dev)
    g = torch.Generator().manual_seed(1)
    big.buf.copy_(torch.randn(big.buf.numel(), generator=g).to(dev).bfloat16())
    small.buf.copy_(torch.randn(small.buf.numel(), generator=g).to(dev).bfloat16())
    conv = K.Conv(big, small, k, k, s, pt, pt)
    conv.pack((torch.randn(k, k, cin, cout, generator=g) * 0.05).to(dev))
    for name in ('fwd', 'bwd_data'):
        outs = {}
        for mode in ('0', '1'):
            os.environ['TDG_PATCH'] = mode
            res = []
            for r in range(reps):
                if name == 'fwd':
                    o = small.like(); conv.fwd(big.ptr(), o.ptr(), n)
                else:
                    o = big.like(); conv.bwd_data(small.ptr(), o.ptr(), n)
                torch.cuda.synchronize()
                res.append(o.buf.float().clone())
            outs[mode] = res
            print(name, 'mode', mode, L.load().tdg_last_kernel().decode(), 'run-to-run identical:', all(torch.equal(res[0], r) for r in res[1:]))
        a, b = outs['0'][0], outs['1'][0]
        d = (a - b).abs()
        shape = (n, oh, ow, small.cs) if name == 'fwd' else (n, h, w, big.cs)
        bad = (d > 0.05 * a.abs().max()).reshape(shape)
        print(name, 'max diff', float(d.max()), 'of', float(a.abs().max()), 'bad elements', int(bad.sum()), 'of', bad.numel())
        if bad.any():
            idx = bad.nonzero()
            print(' bad images', torch.unique(idx[:, 0])[:40].tolist(), '... count', len(torch.unique(idx[:, 0])))
            print(' bad rows', torch.unique(idx[:, 1]).tolist(), 'cols', torch.unique(idx[:, 2]).tolist())
            ch = torch.unique(idx[:, 3])
            print(' bad channels: count', len(ch), 'min', int(ch.min()), 'max', int(ch.max()), ch[:24].tolist())

if __name__ == '__main__':
    nums = [int(a) for a in sys.argv[1:]]
    run(*(nums if len(nums) == 7 else (1536, 8, 8, 400, 800, 5, 2)))
