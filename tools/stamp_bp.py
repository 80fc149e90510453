"""Diagnostics: prologue / K loop / epilogue of igemm_fwd_bp_kernel per wave (in-kernel stamps, shader clocks).
Needs the diagnostic library: `bash 3dgan_amd/csrc/build.sh stamps`.   usage: python tools/stamp_bp.py n h w cin cout"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['TDG_LIB_PATH'] = os.path.join(ROOT, '3dgan_amd', 'lib3dgan_hip_stamps.so')
os.environ['TDG_BLOCKPATCH'] = '1'
sys.path.insert(0, ROOT)
import torch
K = importlib.import_module('3dgan_amd.kernels')
L = importlib.import_module('3dgan_amd._lib')


def main():
    n, h, w, cin, cout = [int(v) for v in sys.argv[1:6]]
    dev = torch.device('cuda:0')
    big, small = K.Act(n, h, w, cin, K.BF16, dev), K.Act(n, h // 2, w // 2, cout, K.BF16, dev)
    big.buf.copy_(torch.randn_like(big.buf.float()).to(big.buf.dtype))
    conv = K.Conv(big, small, 4, 4, 2, 1, 1)
    conv.pack(torch.randn(4, 4, cin, cout, device=dev) * 0.05)
    stamps = torch.zeros(2 * 262144, dtype=torch.int64, device=dev)
    os.environ['TDG_STAMP_PTR'] = str(stamps.data_ptr())
    for _ in range(3):
        conv.fwd(big.ptr(), small.ptr(), n)
    torch.cuda.synchronize()
    print(L.load().tdg_last_kernel().decode())
    w = stamps[262144:].cpu().view(-1, 8, 4).double()
    st = stamps[:262144].cpu().view(-1, 8, 4).double()
    w = w[st[:, 0, 0] > 0]
    st = st[st[:, 0, 0] > 0]
    nsteps = 16 * cin // 64
    for i, nm in enumerate(('prologue', 'K loop', 'epilogue')):
        d = st[:, :, i + 1] - st[:, :, i]
        print('%-9s mean %8.0f  min %8.0f  max %8.0f clocks%s' % (nm, d.mean(), d.min(), d.max(), '  (%.0f per step over %d steps)' % (d.mean() / nsteps, nsteps) if i == 1 else ''))
    print('of the K loop, per step: waiting for own pieces (vmcnt) %.0f, at the barrier %.0f clocks' % (w[:, :, 0].mean() / nsteps, w[:, :, 1].mean() / nsteps))
    print('workgroups', st.shape[0])


if __name__ == '__main__':
    main()
