"""Diagnostics: phase lengths of thin_fwd_kernel per workgroup (in-kernel s_memtime stamps, 100 MHz ticks).
Needs the diagnostic library: `bash 3dgan_amd/csrc/build.sh stamps`.
usage: python tools/stamp_thin.py [n_images]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['TDG_LIB_PATH'] = os.path.join(ROOT, '3dgan_amd', 'lib3dgan_hip_stamps.so')
sys.path.insert(0, ROOT)
import torch
K = importlib.import_module('3dgan_amd.kernels')


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
    h = w = 32
    dev = torch.device('cuda:0')
    big, small = K.Act(n, h, w, 3, K.BF16, dev), K.Act(n, 16, 16, 200, K.BF16, dev)
    big.buf.copy_(torch.randn_like(big.buf.float()).to(big.buf.dtype))
    conv = K.Conv(big, small, 5, 5, 2, 1, 1)
    conv.pack(torch.randn(5, 5, 3, 200, device=dev) * 0.05)
    bias = torch.zeros(200, device=dev)
    nwg = n * 2
    stamps = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
    os.environ['TDG_STAMP_PTR'] = str(stamps.data_ptr())
    epi = K.epilogue(bias=bias, act=K.ACT_LRELU, leak=0.2)
    for _ in range(3):
        conv.fwd(big.ptr(), small.ptr(), n, epi)
    torch.cuda.synchronize()
    st = stamps.cpu().view(nwg, 4, 8).double()
    t = st[:, :, :5]
    t0 = t[:, :, 0].min()
    names = ('stage', 'mma', 'epi->lds', 'store')
    for i, nm in enumerate(names):
        d = t[:, :, i + 1] - t[:, :, i]
        print('%-9s mean %7.1f  min %7.1f  max %7.1f ticks' % (nm, d.mean(), d.min(), d.max()))
    life = t[:, :, 4] - t[:, :, 0]
    print('lifetime  mean %7.1f  (kernel span %.1f ticks = %.1f us)' % (life.mean(), t[:, :, 4].max() - t0, (t[:, :, 4].max() - t0) / 100))
    # concurrency: workgroups alive at the kernel's midpoint
    mid = t0 + (t[:, :, 4].max() - t0) / 2
    alive = ((t[:, 0, 0] <= mid) & (t[:, 0, 4] >= mid)).sum().item()
    print('workgroups alive at mid-kernel: %d' % alive)
    hw = st[:, 0, 5].long()
    cu = (hw >> 8) & 0xf
    se = (hw >> 13) & 0x7
    print('distinct (se,cu) ids seen: %d' % len(set(zip(se.tolist(), cu.tolist()))))


if __name__ == '__main__':
    main()
