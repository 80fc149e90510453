"""Diagnostics: where a K step of the LDS-DMA GEMM spends its cycles (in-kernel s_memtime stamps).
Needs the diagnostic library: `bash 3dgan_amd/csrc/build.sh stamps`.  Read SHARES, not lengths.
usage: python tools/stamp_conv.py n h w cin cout k stride [fwd|bwd_data]"""
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
    which = sys.argv[8] if len(sys.argv) > 8 else 'fwd'
    dev = torch.device('cuda:0')
    oh, ow = -(-h // s), -(-w // s)
    pt = max((oh - 1) * s + k - h, 0) // 2
    big, small = K.Act(n, h, w, cin, K.BF16, dev), K.Act(n, oh, ow, cout, K.BF16, dev)
    big.buf.copy_(torch.randn_like(big.buf.float()).to(big.buf.dtype))
    small.buf.copy_(torch.randn_like(small.buf.float()).to(small.buf.dtype))
    conv = K.Conv(big, small, k, k, s, pt, pt)
    conv.pack(torch.randn(k, k, cin, cout, device=dev) * 0.05)
    out = big.like()
    stamps = torch.zeros(2 * 8192 * 8 * 4, dtype=torch.int64, device=dev)
    os.environ['TDG_STAMP_PTR'] = str(stamps.data_ptr())
    fn = (lambda: conv.fwd(big.ptr(), small.ptr(), n)) if which == 'fwd' else (lambda: conv.bwd_data(small.ptr(), out.ptr(), n))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ph = stamps[262144:].cpu().view(-1, 8, 4).double()
    st = stamps[:262144].cpu().view(-1, 8, 4)
    st = st[st[:, 0, 3] > 0].double()
    steps = st[:, :, 3]
    for i, name in enumerate(('issue', 'mma', 'barrier')):
        per = st[:, :, i] / steps
        print('%-8s cycles/step (100 MHz ticks x?): mean %.1f  min %.1f  max %.1f' % (name, per.mean(), per.min(), per.max()))
    tot = (st[:, :, 0] + st[:, :, 1] + st[:, :, 2]) / steps
    print('total    %.1f ticks/step over %d blocks' % (tot.mean(), st.shape[0]))
    ph = ph[ph[:, 0, 0] > 0]
    if ph.shape[0]:
        life = ph[:, :, 3] - ph[:, :, 0]
        for nm, d in (('prologue + K loop', ph[:, :, 1] - ph[:, :, 0]), ('epilogue -> LDS', ph[:, :, 2] - ph[:, :, 1]),
                      ('LDS -> global', ph[:, :, 3] - ph[:, :, 2])):
            print('%-18s %5.1f %% of the workgroup lifetime (mean %.0f ticks)' % (nm, 100 * (d / life).mean(), d.mean()))
    if ph.shape[0] and ph.shape[0] == st.shape[0]:
        # prologue of each workgroup = (kernel entry -> end of the K loop) minus the loop's own stamped cycles (wave 0), by
        # start order: the first 256 workgroups are the launch's first round (every CU in its prologue at once)
        pro = (ph[:, 0, 1] - ph[:, 0, 0]) - (st[:, 0, 0] + st[:, 0, 1] + st[:, 0, 2])
        order = torch.argsort(ph[:, 0, 0])
        pro = pro[order]
        q = lambda t: ' '.join('%.0f' % v for v in torch.quantile(t, torch.tensor([0.05, 0.5, 0.95], dtype=t.dtype)).tolist())
        print('prologue cycles (5 / 50 / 95 %%): first 256 workgroups %s | the rest %s' % (q(pro[:256]), q(pro[256:]) if pro.numel() > 256 else '-'))
        t0 = ph[order, 0, 0]
        print('start spread of the first 256: %.0f cycles; of all: %.0f' % ((t0[255] - t0[0]).item() if t0.numel() > 255 else -1, (t0[-1] - t0[0]).item()))
    per_wave = (st[:, :, :3] / steps[:, :, None]).mean(0)
    for wv in range(8):
        print('wave %d: issue %.1f mma %.1f barrier %.1f' % (wv, *per_wave[wv].tolist()))


if __name__ == '__main__':
    main()
