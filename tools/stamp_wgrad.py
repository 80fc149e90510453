"""Diagnostics: phase lengths of igemm_wgrad_dma_kernel per wave (in-kernel s_memtime stamps).
Needs the diagnostic library: `bash 3dgan_amd/csrc/build.sh stamps`.  Read SHARES, not lengths.
usage: python tools/stamp_wgrad.py n h w cin cout k stride"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['TDG_LIB_PATH'] = os.path.join(ROOT, '3dgan_amd', 'lib3dgan_hip_stamps.so')
sys.path.insert(0, ROOT)
import torch
K = importlib.import_module('3dgan_amd.kernels')


def main():
    n, h, w, cin, cout, k, s = [int(v) for v in sys.argv[1:8]]
    dev = torch.device('cuda:0')
    oh, ow = -(-h // s), -(-w // s)
    pt = max((oh - 1) * s + k - h, 0) // 2
    big, small = K.Act(n, h, w, cin, K.BF16, dev), K.Act(n, oh, ow, cout, K.BF16, dev)
    big.buf.copy_(torch.randn_like(big.buf.float()).to(big.buf.dtype))
    small.buf.copy_(torch.randn_like(small.buf.float()).to(small.buf.dtype))
    conv = K.Conv(big, small, k, k, s, pt, pt)
    dw = torch.zeros(k, k, cin, cout, device=dev)
    stamps = torch.zeros(4096 * 8 * 8, dtype=torch.int64, device=dev)
    os.environ['TDG_STAMP_PTR'] = str(stamps.data_ptr())
    for _ in range(3):
        conv.bwd_filter(big.ptr(), small.ptr(), dw, n)
    torch.cuda.synchronize()
    st = stamps.cpu().view(-1, 8, 8).double()
    st = st[st[:, 0, 0] > 0]
    pro, loop, epi = st[:, :, 1] - st[:, :, 0], st[:, :, 2] - st[:, :, 1], st[:, :, 3] - st[:, :, 2]
    life = st[:, :, 3] - st[:, :, 0]
    print('%d workgroups; lifetime mean %.0f ticks' % (st.shape[0], life.mean()))
    for nm, d in (('prologue+first load', pro), ('loop', loop), ('epilogue', epi)):
        print('%-20s %5.1f %%  (mean %.0f, max %.0f)' % (nm, 100 * (d / life).mean(), d.mean(), d.max()))
    for nm, i in (('  loop: mma+issue', 4), ('  loop: vmcnt wait', 5), ('  loop: barrier', 6)):
        print('%-20s %5.1f %% of the loop' % (nm, 100 * (st[:, :, i] / loop).mean()))
    span = st[:, :, 3].max() - st[:, :, 0].min()
    print('kernel span %.0f ticks; workgroup lifetime / span = %.2f' % (span, life.mean() / span))


if __name__ == '__main__':
    main()
