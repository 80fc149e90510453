"""Diagnostics: phase lengths of bwd_fused_kernel per wave (in-kernel s_memtime stamps, shader clocks).
Needs the diagnostic library: `bash 3dgan_amd/csrc/build.sh stamps`.   usage: python tools/stamp_fused.py [n_images]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['TDG_LIB_PATH'] = os.path.join(ROOT, '3dgan_amd', 'lib3dgan_hip_stamps.so')
sys.path.insert(0, ROOT)
import torch
K = importlib.import_module('3dgan_amd.kernels')


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dev = torch.device('cuda:0')
    big, small = K.Act(n, 32, 32, 3, K.BF16, dev), K.Act(n, 16, 16, 200, K.BF16, dev)
    small.buf.copy_(torch.randn_like(small.buf.float()).to(small.buf.dtype))
    conv = K.Conv(big, small, 5, 5, 2, 1, 1)
    conv.pack(torch.randn(5, 5, 3, 200, device=dev) * 0.05)
    nwg = n * 4
    stamps = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device=dev)
    os.environ['TDG_STAMP_PTR'] = str(stamps.data_ptr())
    for _ in range(3):
        conv.bwd_data(small.ptr(), big.ptr(), n)
    torch.cuda.synchronize()
    st = stamps.cpu().view(nwg, 8, 8).double()
    used = st[:, 0, 0] > 0
    t = st[used][:, :, :5]
    print('workgroups:', int(used.sum()))
    work = t[:, :, 3] > 0                                   # waves that multiplied (a small tile leaves some without a pixel tile)
    print('waves with a pixel tile: %d of %d' % (int(work.sum()), work.numel()))
    for i, nm in enumerate(('issue loads -> landed (vmcnt)', 'barrier', 'multiply loop', 'stores (incl. vmcnt(0))')):
        d = (t[:, :, i + 1] - t[:, :, i])[work]
        print('%-32s mean %8.0f  min %8.0f  max %8.0f clocks' % (nm, d.mean(), d.min(), d.max()))
    wg = t[:, :, 4].max(dim=1).values - t[:, :, 0].min(dim=1).values
    print('workgroup lifetime mean %.0f  min %.0f  max %.0f clocks' % (wg.mean(), wg.min(), wg.max()))
    life = t[:, :, 4] - t[:, :, 0]
    span = t[:, :, 4].max() - t[:, :, 0].min()
    print('wave lifetime mean %.0f clocks; kernel span %.0f clocks' % (life.mean(), span))


if __name__ == '__main__':
    main()
