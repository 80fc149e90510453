"""Diagnostics: where does igemm_wgrad_patch_kernel differ from the oracle?  usage: python tools/dbg_wpatch.py n h w cin cout k s"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
K = importlib.import_module('3dgan_amd.kernels')
from oracle import tf_ops as T

def main(n, h, w, cin, cout, k, s):
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(11)
    rb = lambda a: torch.tensor(a, dtype=torch.float32).bfloat16().float().numpy()
    x = rb(rng.standard_normal((n, h, w, cin)).astype(np.float32))
    oh, pt, _ = T.same_pad(h, k, s); ow, pl, _ = T.same_pad(w, k, s)
    big, small = K.Act(n, h, w, cin, K.BF16, dev), K.Act(n, oh, ow, cout, K.BF16, dev)
    conv = K.Conv(big, small, k, k, s, pt, pl)
    dy = rb(rng.standard_normal((n, oh, ow, cout)).astype(np.float32))
    big.set(x); small.set(dy)
    ref = T.conv2d_backprop_filter(x.astype(np.float64), (k, k, cin, cout), dy.astype(np.float64), s)
    dw = torch.zeros((k, k, cin, cout), device=dev)
    conv.bwd_filter(big.ptr(), small.ptr(), dw, n)
    torch.cuda.synchronize()
    lib = importlib.import_module('3dgan_amd._lib').load()
    print(lib.tdg_last_kernel().decode())
    got = dw.cpu().numpy()
    err = np.abs(got - ref) / (np.abs(ref).max() + 1e-30)
    err = np.where(np.isfinite(err), err, 9.0)
    print('max rel err', err.max())
    bad = err > 2e-2
    print('bad per tap (kh x kw):'); print(bad.reshape(k, k, -1).sum(-1))
    print('bad per channel:', bad.sum((0, 1, 3)))
    bn = bad.sum((0, 1, 2))
    print('bad per 16-column tile:', bn.reshape(-1, 16).sum(1) if cout % 16 == 0 else bn)

if __name__ == '__main__':
    main(*[int(a) for a in sys.argv[1:8]])
