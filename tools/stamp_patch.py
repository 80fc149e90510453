"""Diagnostics: phases of igemm_fwd_patch_kernel's workgroups (stamps library): prologue, K loop, epilogue in cycles."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['TDG_LIB_PATH'] = os.path.join(ROOT, '3dgan_amd', 'lib3dgan_hip_stamps.so')
sys.path.insert(0, ROOT)
import torch
K = importlib.import_module('3dgan_amd.kernels')
n, h, w, cin, cout, k, s = [int(v) for v in sys.argv[1:8]]
which = sys.argv[8] if len(sys.argv) > 8 else 'fwd'
dev = torch.device('cuda:0')
oh, ow = -(-h // s), -(-w // s)
pt = max((oh - 1) * s + k - h, 0) // 2
big, small = K.Act(n, h, w, cin, K.BF16, dev), K.Act(n, oh, ow, cout, K.BF16, dev)
big.buf.copy_(torch.randn_like(big.buf.float()).to(big.buf.dtype)); small.buf.copy_(torch.randn_like(small.buf.float()).to(small.buf.dtype))
conv = K.Conv(big, small, k, k, s, pt, pt)
conv.pack(torch.randn(k, k, cin, cout, device=dev) * 0.05)
out = big.like()
stamps = torch.zeros(3 * 8192 * 8 * 4, dtype=torch.int64, device=dev)
os.environ['TDG_STAMP_PTR'] = str(stamps.data_ptr())
fn = (lambda: conv.fwd(big.ptr(), small.ptr(), n)) if which == 'fwd' else (lambda: conv.bwd_data(small.ptr(), out.ptr(), n))
for _ in range(3):
    fn()
torch.cuda.synchronize()
ph = stamps[262144:524288].cpu().view(-1, 8, 4).double()
qq = stamps[524288:].cpu().view(-1, 8, 4).double()
ph = ph[ph[:, 0, 0] > 0]
st = stamps[:262144].cpu().view(-1, 8, 4).double(); st = st[st[:, 0, 3] > 0]
for wv, nm in ((0, 'compute wave 0'), (4, 'loader wave 4')):
    pro, loop, epi = ph[:, wv, 1] - ph[:, wv, 0], ph[:, wv, 2] - ph[:, wv, 1], ph[:, wv, 3] - ph[:, wv, 2]
    tot = ph[:, wv, 3] - ph[:, wv, 0]
    print('%s: prologue %.0f (%.1f %%)  K loop %.0f (%.1f %%, %.0f per step incl. stamp cost)  epilogue %.0f (%.1f %%)  lifetime %.0f cycles, %d workgroups' % (
        nm, pro.mean(), 100 * (pro / tot).mean(), loop.mean(), 100 * (loop / tot).mean(), (loop / st[:, wv, 3]).mean(), epi.mean(), 100 * (epi / tot).mean(), tot.mean(), ph.shape[0]))
order = torch.argsort(ph[:, 0, 0])
pro = (ph[:, 0, 1] - ph[:, 0, 0])[order]
print('prologue of the first 256 workgroups %.0f, of the rest %.0f' % (pro[:256].mean(), pro[256:].mean() if pro.numel() > 256 else -1))
for wv in range(8):
    print('wave %d: barrier wait %.0f per step%s' % (wv, (st[:, wv, 2] / st[:, wv, 3]).mean(), '' if wv < 4 else ', piece issue %.0f per step' % (st[:, wv, 0] / st[:, wv, 3]).mean()))
print('compute wave 0: entry -> tables built %.0f cycles; loader wave 4: entry -> prologue pieces issued %.0f cycles' % (st[:, 0, 0].mean(), st[:, 4, 1].mean()))
qq = qq[:ph.shape[0]] if qq.shape[0] >= ph.shape[0] else qq
m = qq[qq[:, 0, 3] > 0]
print('compute wave 0, cycles since entry: setup done %.0f, taps in lanes %.0f, chunk table written %.0f, bias row written %.0f' % tuple(m[:, 0, i].mean() for i in range(4)))
